// device.hip — Device / SceneGPU / ProbeGPU / Renderer behind the C ABI (include/lpt.h).
//
// Host orchestration of one frame follows Renderer::raytrace
// (reference crates/lib/src/renderer.rs:392-549): pass order, the seed / bounces /
// frame_count protocol and the accumulate flag.  raytrace() RECORDS (the reference records into an encoder the app submits
// once, app.rs:335-337); a submission point launches the recorded calls as wavefronts of several samples per pixel
// (record_call / flush_pending / wavefront_trace + wavefront_finish).  A renderer enqueues on its own HIP stream (accumulation,
// filter passes, reads, the frame exchange: in call order) and on the streams of its wavefront lanes (the
// traversal / shading of consecutive wavefronts, overlapped); nothing here waits for the GPU except the
// read-back calls (the reference's only blocking point is read_pixels, :791) and scene edits.
// Also here: the multi-GPU frame exchange (plain RCCL, opened with dlopen: lpt_comm_*, lpt_renderer_exchange) and the
// tile-ownership rule (weighted shards).
#include <hip/hip_runtime.h>

#include <dlfcn.h>

#include <algorithm>
#include <cmath>
#include <map>
#include <cstdlib>
#include <mutex>
#include <new>

#include "common.h"
#include "kernels.h"
#include "build_kernels.h"
#include <hipcub/hipcub.hpp>
#include <rccl/rccl.h>

using namespace lpt;
using namespace lptd;

#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t e__ = (expr);                                                                   \
        if (e__ != hipSuccess) return fail(LPT_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e__)); \
    } while (0)

struct lpt_device {
    int ordinal = 0;
    hipStream_t stream = nullptr;
    int compute_units = 0;
    char name[128] = {0};
    std::vector<lpt_renderer *> renderers;   // live renderers of this device: scene / probe edits submit their recorded calls first
};

// one RCCL communicator rank (frame exchange); created by lpt_comm_create
struct lpt_comm {
    lpt_device *dev = nullptr;
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1;
};

// librccl is opened on the first lpt_comm_* call, not at load time: a single-GPU host never needs it (and never pays for
// loading it).  rccl.h supplies the types only; nothing here links against the library.
struct Rccl {
    void *handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int *) = nullptr;
    ncclResult_t (*CommUserRank)(const ncclComm_t, int *) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Reduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, int, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
};
// nullptr (and lpt_last_error set) when librccl cannot be opened
static const Rccl *rccl() {
    static Rccl table;
    static int state = 0;   // 0 = not tried, 1 = loaded, -1 = failed
    static std::string why = "missing symbol";   // dlerror() of the failed attempt (it reads once: a second call returns NULL)
    static std::mutex mu;
    std::lock_guard<std::mutex> lock(mu);
    if (state == 1) return &table;
    if (state == 0) {
        const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
        std::string rocm_path;
        if (const char *rp = getenv("ROCM_PATH")) rocm_path = std::string(rp) + "/lib/librccl.so";
        void *h = nullptr;
        // LPT_RCCL_LIBRARY: another build of RCCL — or the multi-process test stand-in (tests/tools/fake_rccl.c), which lets N
        // ranks share the one GPU of a test box; when it is set nothing else is tried
        state = -1;
        if (const char *lp = getenv("LPT_RCCL_LIBRARY")) { h = dlopen(lp, RTLD_NOW | RTLD_LOCAL); rocm_path.clear(); if (!h) { if (const char *e = dlerror()) why = e; goto rccl_done; } }
        if (!rocm_path.empty()) h = dlopen(rocm_path.c_str(), RTLD_NOW | RTLD_LOCAL);
        for (size_t i = 0; !h && i < sizeof names / sizeof names[0]; ++i) h = dlopen(names[i], RTLD_NOW | RTLD_LOCAL);
        if (!h) { if (const char *e = dlerror()) why = e; }
        if (h) {
            table.handle = h;
            bool ok = true;
#define RCCL_SYM(field, sym) do { *(void **)(&table.field) = dlsym(h, sym); if (!table.field) ok = false; } while (0)
            RCCL_SYM(GetUniqueId, "ncclGetUniqueId"); RCCL_SYM(CommInitRank, "ncclCommInitRank"); RCCL_SYM(CommDestroy, "ncclCommDestroy");
            RCCL_SYM(CommCount, "ncclCommCount"); RCCL_SYM(CommUserRank, "ncclCommUserRank");
            RCCL_SYM(GroupStart, "ncclGroupStart"); RCCL_SYM(GroupEnd, "ncclGroupEnd"); RCCL_SYM(Send, "ncclSend"); RCCL_SYM(Recv, "ncclRecv");
            RCCL_SYM(Reduce, "ncclReduce"); RCCL_SYM(GetErrorString, "ncclGetErrorString");
#undef RCCL_SYM
            if (ok) state = 1;
        }
    rccl_done:;
    }
    if (state == 1) return &table;
    fail(LPT_ERR_RCCL, "librccl could not be loaded (%s): the multi-GPU frame exchange needs RCCL", why.c_str());
    return nullptr;
}

#define RCCL_TRY(expr)                                                                                \
    do {                                                                                              \
        ncclResult_t r__ = (expr);                                                                    \
        if (r__ != ncclSuccess) return fail(LPT_ERR_RCCL, "%s failed: %s", #expr, rccl() ? rccl()->GetErrorString(r__) : "?"); \
    } while (0)

struct lpt_scene_gpu {
    lpt_device *dev = nullptr;
    DScene d{};
    float max_abs = 0.f;            // the scene's largest |coordinate|: d.pad_abs = kScenePad x this (host build: Accel::max_abs; GPU build: the bounds pass; refit: the root box)
    void *nodes = nullptr, *woop = nullptr, *leaf_prim = nullptr, *tri_verts = nullptr;
    void *materials = nullptr, *lights = nullptr, *texels = nullptr, *images = nullptr, *srgb_lut = nullptr;
    // paired textures (kernels.h DScene::pair_texels): (albedo image, mra image) -> pair index, for the materials whose two
    // textures have one size; the shading records of such materials carry kPairedBit | pair index instead of two image ids
    void *pair_texels = nullptr, *pair_images = nullptr;
    std::map<std::pair<uint32_t, uint32_t>, uint32_t> pair_map;
    std::vector<DImage> pair_descs;   // host copy: a paired material's record carries its pair's offset and size directly
    // which images the tiled atlas holds on their own: an image that only ever appears as half of a pair is not stored a second
    // time (scene.rs:172-184 keeps each image once)
    std::vector<uint8_t> image_resident;
    lpt_accel_stats stats{};
    // refit bookkeeping (lpt_scene_gpu_update_instances)
    void *node_lo = nullptr, *node_hi = nullptr;  // per-node world box
    void *obj_verts = nullptr, *obj_indices = nullptr, *bad_flag = nullptr;  // object-space meshes for device-side re-baking
    void *arena = nullptr;                                                    // scratch of the GPU builder, kept for rebuilds
    size_t arena_bytes = 0;
    std::vector<uint32_t> inst_first, inst_count, level_start;
    std::vector<lpt_instance> instances;                                // as baked
    size_t n_entries = 0, n_vertices = 0, n_indices = 0;
};

struct lpt_probe {
    lpt_device *dev = nullptr;
    DProbe d{};
    void *rgbe = nullptr;
};

enum { ST_RAYGEN = 0, ST_INTERSECT, ST_SHADE, ST_SHADOW, ST_ACCUM, ST_ASVGF, ST_EXCHANGE, ST_PRIMARY, ST_PATH, ST_COUNT };
// "primary intersection": the IntersectorPass of bounce 0 when it runs as packet traversal (k_trace_packet), timed apart from the
// per-ray traversal launches ("intersection") because it is another kernel
// "path": every bounce behind the primary hits in ONE launch (k_path), the form small wavefronts — the tile shard of a multi-GPU frame — take
static const char *kStageLabel[ST_COUNT] = {"ray generation", "intersection", "shading", "shadow", "accumulation", "asvgf", "exchange", "primary intersection", "path"};

// One independent wavefront context ("lane") of a renderer: everything a raytrace() call owns while its rays are in flight.
// Consecutive raytrace() calls of ONE renderer take the lanes in turn (default 2), so the traversal / shading of call k+1
// overlaps the drain tails of call k — what several renderers do for several frames, for the unchanged caller that issues
// its samples one raytrace() at a time.  Accumulation stays on the renderer's stream, in call order (the fp32 sums are
// order-sensitive), so results do not depend on the number of lanes.
struct Wavefront {
    hipStream_t stream = nullptr;   // the lane's own stream (unused with one lane: everything runs on the renderer's)
    Queue q[2]{};
    ShadowQueue sq{};
    float4 *hits = nullptr, *Lsum = nullptr;
    uint32_t *strag = nullptr;      // the stragglers of a traversal launch (k_trace's step budget -> k_trace_coop): ray index | shadow << 31
    FrameCounters *ctr = nullptr;
    size_t ray_cap = 0;             // rays (pixel slots x samples) the per-ray buffers can hold; 0 = not allocated yet
    hipEvent_t done = nullptr;      // recorded on `stream` behind the lane's last traversal / shading launch
    hipEvent_t consumed = nullptr;  // recorded on the renderer's stream behind the accumulation that read this lane's Lsum
    bool consumed_recorded = false;
};
constexpr int kMaxLanes = 4;
constexpr uint32_t kStepBudget = 48u, kBudgetRays = 3000000u;  // defaults of LPT_EXP_STEP_BUDGET / LPT_EXP_BUDGET_RAYS (measured: profiles/r04_experiments_ab.txt H)
constexpr float kPacketMaxPixelRad = 1.8e-3f;    // bounce 0 as packets up to this angle per pixel (measured: 1.53 mrad, 960x540: packets 4.46 against 4.51 ms; 2.05 mrad, 720x405: 3.32 against 3.20)
constexpr uint32_t kOccEntries = 1u << 18;       // occluder-cache probe: 1 MB, L2-resident
constexpr uint32_t kPathRays = 120000u;           // rays of a wavefront up to which the path kernel is used: the measured cross-over against the per-bounce launches with their tails in place
                                                  // (whole frames, ms: 113 k rays 0.913 path / 0.932 per bounce, 147 k 1.07 / 0.99, 332 k 1.89 / 1.40; it was 450 000 against the budget pair)
constexpr uint64_t kSplitRays = 3000000ull;       // a batch above this leaves as at least two wavefronts (LPT_EXP_SPLIT_RAYS)
constexpr uint32_t kCoopRays = 32000u;            // rays of a wavefront up to which EVERY ray is traced by a whole wave (k_trace_coop over the queues): the measured cross-over (28 k rays:
                                                  // 0.755 -> 0.652 ms per frame, 37 k: 0.765 -> 0.787; profiles/r05_experiments_ab.txt V)
constexpr uint32_t kPacketBlocksPerCu = 128u;   // k_trace_packet's grid: one-wave blocks, packets dealt by stride (32 / 64 per CU: the same, round 5)
constexpr uint32_t kCoopWavesPerCu = 32u;   // k_trace_coop's grid: a wave per straggler, most waves find none and leave (8 / 4 per CU: the same 2.90 ms per 1/8-shard frame, round 5)
constexpr uint64_t kWavefrontRays = 1ull << 22;   // rays (pixel slots x samples) per wavefront an automatic submission aims at

struct lpt_renderer {
    lpt_device *dev = nullptr;
    hipStream_t stream = nullptr;   // every renderer enqueues on its OWN stream, so two renderers pipeline
                                    // consecutive frames (and a frame's collective overlaps the next frame)
    Wavefront wf[kMaxLanes];
    int n_lanes = 2;                // lpt_renderer_set_lanes
    uint32_t lane_rr = 0;
    int last_lane = 0;
    uint32_t req_w = 0, req_h = 0, w = 0, h = 0;
    float downsample = 0.5f;
    const lpt_scene_gpu *sg = nullptr;
    const lpt_probe *probe = nullptr;
    bool resources_set = false;
    // global_uniforms (renderer.rs:286-290)
    uint32_t frame_count = 1, seed = 0;
    bool accumulate = false, frame_back = true;
    // Recorded, not yet submitted raytrace() calls ("record now, submit later": the reference records every pass of a frame
    // into one encoder, renderer.rs:392-549, and the app submits it once, app.rs:335-337).  Consecutive calls with the same
    // view that continue one accumulation fuse into ONE wavefront of `n` samples per pixel; the host-side protocol state
    // (frame_count, seed, frame_back) moves at record time, the snapshot below is what the launches use.
    struct Pending { float view[16]; uint32_t n = 0, frame_count0 = 1, seed0 = 0; bool acc0 = false; } pend;
    uint32_t max_fused = 0;    // 0 = auto: up to 64 calls wait for the next submission point, which cuts them into wavefronts of about 4 M rays
                               // (spatially: runs of tile rows x all the samples); n >= 1: n calls are ONE wavefront and launch when the n-th is recorded
    uint32_t packet_primary = 2u;  // bounce 0 by packet traversal (k_trace_packet): 2 = where an 8x8-pixel patch is narrow enough (wavefront_trace), 1 = always, 0 = never (LPT_OPT_PACKET_PRIMARY)
    uint32_t pipe_rays = 0x7FFFFFFFu;   // wavefronts of at most this many rays trace with the one-round-trip step (k_trace<.., PIPE>): all of them; LPT_EXP_PIPE_RAYS 0: none
    uint64_t wavefront_rays = kWavefrontRays;   // LPT_OPT_WAVEFRONT_RAYS: tests cut small frames into many wavefronts
    // wavefronts of at most this many rays run every bounce behind the primary hits in ONE launch (k_path: no chip-wide barrier per
    // bounce); larger ones take the per-bounce launches, whose drains are then a few per cent.  LPT_OPT_PATH_RAYS; 0: never
    uint32_t path_rays = kPathRays;
    uint32_t path_waves_per_cu = 16;   // k_path: 4 waves per SIMD (kernels.h LPT_PATH_ATTR)
    int path_refill = 32;              // k_path: a batch of lanes is shaded (and idle lanes start new paths) when at most this many lanes are tracing
    uint64_t n_recorded = 0, n_wavefronts = 0;   // raytrace() calls recorded / wavefronts submitted so far (lpt_renderer_get_submission_stats)
    int mode = LPT_BLIT_PATHTRACE;
    // build-only knobs
    uint32_t max_bounces = 3, user_seed = 0;
    float vfov = 0.78539816339744830962f;
    uint32_t rank = 0, world = 1, tile_w = 32, tile_h = 8;
    std::vector<uint32_t> weights;   // tile-ownership weights of the ranks (empty = every rank 1: tile id mod world)
    ShardMap map{1u, 1u, {0u}};      // this rank's part of the ownership rule
    ShardTable h_table{};            // rank 0: the whole rule ...
    std::vector<uint32_t> h_offset;  // ... the staging offsets of the ranks (world + 1 entries, any world size) ...
    ShardTable *d_table = nullptr;   // ... and the rule's copy in device memory for the unpack kernels; refreshed by alloc_frame_buffers
    bool use_noise = false, stats = false, timings = false;
    // traversal tuning (lpt_renderer_set_option, for experiments)
    int refill = 44;
    int sort_queues = 0;       // k_shade emits both ray queues ordered by direction octant within a block (lpt_renderer_set_sort_queues)
    // k_shade grid, blocks per CU (LPT_EXP_SHADE_BLOCKS_PER_CU); 0 = by the submission: 4 (what is resident at 4 waves / SIMD; 8: a 1/8 shard 1.89 instead of 1.80 ms) for a
    // wavefront that has the chip to itself, 3 for the pieces of a cut batch — three 128-VGPR shading waves leave a SIMD room for one 72-VGPR k_trace wave of the piece on the
    // other lane: the kernel alone is 10 % slower (2.75 -> 3.05 ms per frame), the frame 0.5 % faster (11.27-11.29 -> 11.19-11.24; profiles/r06_experiments_ab.txt O)
    uint32_t shade_blocks_per_cu = 0;
    uint32_t trace_waves_per_cu = 0;  // 0 = sized from the frame's ray count (below); LPT_EXP_TRACE_WAVES_PER_CU pins it
    // device memory
    uint32_t n_slots = 0;
    float4 *accum = nullptr, *scratch = nullptr;
    // frame exchange (lpt_renderer_exchange): `frame` = the presented whole frame on rank 0 (accum stays owned-only),
    // `xstage` = packed owned tiles (this rank's; on rank 0 those of every rank, concatenated)
    lpt_comm *comm = nullptr;
    float4 *frame = nullptr, *xstage = nullptr;
    size_t xstage_elems = 0;
    bool presented = false;   // read_radiance / read_pixels / blit resolve from `frame` (set by an exchange, cleared by raytrace)
    hipEvent_t xevent = nullptr;
    bool xevent_recorded = false;
    Totals *totals = nullptr;
    // per-bounce traversal launches: a ray that is not finished after this many steps is handed to k_trace_coop (a whole wave per ray); 0 = off.
    // Applies to wavefronts of at most `budget_rays` rays: where a launch's longest ray sets its duration
    uint32_t step_budget = kStepBudget, budget_rays = kBudgetRays;
    bool budget_split = false;     // the budget also for the pieces of a cut batch (LPT_EXP_BUDGET_SPLIT)
    // the same launches' tails finished IN PLACE instead (kernels.h tail_park / tail_walk): a wave whose queues are dry and that is down to this many live rays finishes them
    // cooperatively from where they stand (default); 0: off (then the step budget + k_trace_coop pair applies).  LPT_OPT_TAIL_LANES
    uint32_t coop_rays = kCoopRays;   // LPT_OPT_COOP_RAYS
    uint32_t tail_lanes = 4u;      // 1/8 shard of the bench frame: 3, 4, 5 the same (2.74 ms per frame against 2.90 with the budget pair), 2 and 8 slower
    uint64_t split_rays = kSplitRays;   // LPT_EXP_SPLIT_RAYS
    uint32_t *err_host = nullptr, *err_dev = nullptr;   // the device's error word: one page-locked host word the kernels write (a bounded wait that ran out), checked behind every blocking call
    bool packet_quads = true;      // a packet of bounce 0 = the four samples of a 4x4-pixel quarter (where the queue order allows it) instead of one sample of an 8x8 patch (LPT_EXP_PACKET_QUADS)
    uint32_t *occ_table = nullptr;   // occluder-cache probe (stats only): kOccEntries leaf slots + 1, zero = empty; allocated by enable_stats
    float occ_cell = 0.25f;          // its grid cell (scene units); LPT_EXP_OCC_CELL_MILLI
    void *default_probe = nullptr;
    float *srgb_thr = nullptr;   // 256 floats: the linear value at which sRGB code i starts (k_tonemap, SPEC §13.2)
    void *noise = nullptr;
    uint32_t noise_w = 0, noise_h = 0;
    // denoiser path (reference render/asvgf.rs ScreenResources :9-152): 2x ping-pong {radiance, gbuffer, moments,
    // history} + motion + radiance_temp; allocated on first use of a denoising BlitMode
    uint4 *den_gbuf[2]{};
    float4 *den_rad[2]{};
    float2 *den_mom[2]{};
    uint32_t *den_hist[2]{};
    float2 *den_motion = nullptr;
    float4 *den_temp = nullptr;
    float4 *den_noisy = nullptr;  // per-pixel sample radiance of the current frame (filter input; exchanged when sharded)
    float4 *den_nd = nullptr;     // the current G-buffer's normal + depth, decoded once per frame for the a-trous passes
    bool den_inputs_ready = false;
    int den_cur = 1;              // current_frame_back starts true (asvgf.rs:233); start() flips it
    CamBasis prev_cam{};          // prev_model_to_screen (renderer.rs:201,319,542-546), identity at start
    // per-stage timing: a ring of event sets (one per raytrace() call) harvested lazily, so
    // timing a run never stalls the host on the frame it has just enqueued
    static constexpr int kRing = 8;
    static constexpr int kMaxEvents = 3 * kMaxBounces + 4;
    hipEvent_t *ev_start = nullptr, *ev_stop = nullptr;  // kRing * kMaxEvents each
    int ev_stage[kRing][kMaxEvents]{};
    int ev_count[kRing]{};
    uint64_t ring_pos = 0;
    double stage_ms[ST_COUNT]{};
    uint32_t stage_launches[ST_COUNT]{};
};

static inline uint32_t div_up(uint32_t a, uint32_t b) { return (a + b - 1u) / b; }
// submits the recorded raytrace() calls; every synchronisation point, and every setter whose value the launches read, runs it first
// what a submission copies to the host as its wavefronts complete (lpt_renderer_read_radiance / lpt_renderer_blit_rgba8)
struct ReadPlan {
    float *radiance = nullptr;      // mean radiance, w*h*4 floats, tight rows
    uint8_t *rgba8 = nullptr;       // or: tonemapped sRGB RGBA8 ...
    size_t row_bytes = 0;           // ... with this row pitch
};
static int flush_pending(lpt_renderer *r, const ReadPlan *read = nullptr);
static void forget_deferred_exchange(lpt_renderer *r);
static FrameParams shard_params(const lpt_renderer *r);
static inline uint32_t stream_grid(const lpt_renderer *r, size_t n);
#define FLUSH_OR_RETURN(r) do { int fst__ = flush_pending(r); if (fst__ != LPT_OK) return fst__; } while (0)
// Recorded raytrace() calls saw the scene / probe as it was when they were issued: an edit (or a destroy) submits them first.
static int flush_device(lpt_device *dev) {
    int st = LPT_OK;
    if (!dev) return st;
    for (lpt_renderer *r : dev->renderers) {
        const int f = flush_pending(r);
        if (st == LPT_OK) st = f;   // the first failure is what the caller reports (lpt_last_error holds its text)
    }
    return st;
}
// what read_radiance / read_pixels / blit show: the exchanged whole frame after lpt_renderer_exchange, else the local target
static inline const float4 *presented_target(const lpt_renderer *r) { return (r->presented && r->frame) ? r->frame : r->accum; }
// The device's error word (a page-locked host word the kernels write when a bounded wait runs out; no shipped kernel has one since k_pool left the library in round 6).  Read behind a
// blocking call, once the stream has been waited for: the frame that raised it is void.
static int check_device_error(lpt_renderer *r) {
    if (!r->err_host) return LPT_OK;
    const uint32_t code = *(volatile uint32_t *)r->err_host;
    if (!code) return LPT_OK;
    *(volatile uint32_t *)r->err_host = 0u;
    return fail(LPT_ERR_HIP, "a device-side bounded wait ran out (code 0x%x): the frame is void", code);
}
static inline size_t stack_bytes(const DScene &sc) { return (size_t)sc.stack_entries * kTraceBlock * sizeof(uint2); }

template <typename T>
static int upload(void **dst, const T *src, size_t count, hipStream_t s) {
    const size_t bytes = sizeof(T) * (count ? count : 1);
    HIP_TRY(hipMalloc(dst, bytes));
    if (count) HIP_TRY(hipMemcpyAsync(*dst, src, sizeof(T) * count, hipMemcpyHostToDevice, s));
    return LPT_OK;
}

// ---------------------------------------------------------------------------- GPU builder (build_kernels.h)
// Fills sg->nodes / woop / leaf_prim / node_lo / node_hi / level_start from the baked triangles that are
// already on the device (sg->d.tri_verts).  `woop_prim` = the Woop maps in prim order (host, SPEC §6).
// `host_woop`: the Woop maps in prim order on the host (upload path) or nullptr when `dev_woop` (device, prim order) is given.
// Scene bounds from the baked triangles on the device (blocking), and what follows from them for every kernel that pads a triangle or writes a node: the scene-wide part
// of the triangle padding (bvh.cpp padded_box; it never shrinks — a pad a little too wide costs nothing) and the scene grid of the node origins (common.h scene_grid, of the
// bounds widened by that padding).  Before a build and before every refit.
static int scene_bounds_and_grid(lpt_scene_gpu *sg, uint32_t n, hipStream_t s, float blo[3], float bhi[3], bool keep_pad) {
    int *db = nullptr, hb[6];
    const int init[6] = {0x7FFFFFFF, 0x7FFFFFFF, 0x7FFFFFFF, (int)0x80000000, (int)0x80000000, (int)0x80000000};
    HIP_TRY(hipMalloc(&db, sizeof init));
    HIP_TRY(hipMemcpyAsync(db, init, sizeof init, hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(k_lbvh_bounds, dim3(std::min<uint32_t>(div_up(n, 256u), 1024u)), dim3(256), 0, s, sg->d, n, db);
    HIP_TRY(hipMemcpyAsync(hb, db, sizeof hb, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    hipFree(db);
    for (int a = 0; a < 6; ++a) { const int o = hb[a] >= 0 ? hb[a] : hb[a] ^ 0x7FFFFFFF; float f; memcpy(&f, &o, 4); (a < 3 ? blo[a] : bhi[a - 3]) = f; }
    float now_abs = 0.0f;
    for (int a = 0; a < 3; ++a) now_abs = std::max(now_abs, std::max(fabsf(blo[a]), fabsf(bhi[a])));
    sg->max_abs = keep_pad ? std::max(sg->max_abs, now_abs) : now_abs;
    sg->d.pad_abs = kScenePad * sg->max_abs;   // (renderers read sg->d at submission)
    // the grid must reach below every PADDED node box: the bounds above are of the bare vertices, and after a refit the kept padding (max_abs never shrinks) may belong
    // to a place farther from the origin than anything is now — widen by the most a triangle is padded by (padded_box: 4e-6 m + pad_abs + 1e-6 extent) before choosing the grid
    float glo[3], ghi[3];
    for (int a = 0; a < 3; ++a) {
        const float e = 4.0f * sg->d.pad_abs + 2e-6f * (bhi[a] - blo[a]) + 1e-30f;
        glo[a] = blo[a] - e;
        ghi[a] = bhi[a] + e;
    }
    scene_grid(glo, ghi, sg->d.grid_lo, sg->d.grid_step);
    return LPT_OK;
}

static int build_lbvh(lpt_scene_gpu *sg, uint32_t n, const WoopTri *host_woop, const float4 *dev_woop, hipStream_t s) {
    const auto t0 = std::chrono::steady_clock::now();
    float blo[3], bhi[3];
    { const int bst = scene_bounds_and_grid(sg, n, s, blo, bhi, false); if (bst != LPT_OK) return bst; }
    float binv[3];
    for (int a = 0; a < 3; ++a) binv[a] = bhi[a] > blo[a] ? 1.0f / (bhi[a] - blo[a]) : 0.0f;

    // scratch: one arena per scene, kept for the next rebuild (about 400 B per triangle); a bump allocator over it
    size_t sort_bytes = 0, scan_bytes = 0;
    hipcub::DeviceRadixSort::SortPairs(nullptr, sort_bytes, (uint32_t *)nullptr, (uint32_t *)nullptr, (uint32_t *)nullptr, (uint32_t *)nullptr, (int)n, 0, 30, s);
    hipcub::DeviceScan::ExclusiveSum(nullptr, scan_bytes, (uint32_t *)nullptr, (uint32_t *)nullptr, (int)n, s);
    const size_t need = (size_t)n * 400u + std::max(sort_bytes, scan_bytes) + 64u * 256u;
    if (sg->arena_bytes < need) {
        if (sg->arena) hipFree(sg->arena);
        sg->arena = nullptr; sg->arena_bytes = 0;
        if (hipMalloc(&sg->arena, need) != hipSuccess) return fail(LPT_ERR_HIP, "GPU BVH build: out of device memory (%zu bytes of scratch)", need);
        sg->arena_bytes = need;
    }
    size_t arena_used = 0;
    auto scratch = [&](size_t bytes) -> void * {
        const size_t at = (arena_used + 255u) & ~(size_t)255u;
        if (at + bytes > sg->arena_bytes) return nullptr;
        arena_used = at + bytes;
        return (char *)sg->arena + at;
    };
    auto done = [&](int st) { hipStreamSynchronize(s); return st; };
#define LB_TRY(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) return done(fail(LPT_ERR_HIP, "GPU BVH build: %s", hipGetErrorString(e_))); } while (0)
#define SCR(type, name, count) type *name = (type *)scratch(sizeof(type) * (size_t)(count)); if (!name) return done(fail(LPT_ERR_HIP, "GPU BVH build: out of device memory"));
    SCR(uint32_t, keys, n) SCR(uint32_t, vals, n) SCR(uint32_t, keys_s, n) SCR(uint32_t, vals_s, n)
    LbvhTree T{};
    SCR(int, t_left, n) SCR(int, t_right, n) SCR(int, t_parent, n) SCR(int, t_leafp, n)
    SCR(uint32_t, t_first, n) SCR(uint32_t, t_last, n) SCR(float4, t_lo, n) SCR(float4, t_hi, n) SCR(float4, t_tlo, n) SCR(float4, t_thi, n) SCR(uint32_t, t_flag, n)
    T.left = t_left; T.right = t_right; T.parent = t_parent; T.leaf_parent = t_leafp; T.first = t_first; T.last = t_last;
    T.lo = t_lo; T.hi = t_hi; T.tri_lo = t_tlo; T.tri_hi = t_thi; T.flag = t_flag; T.sorted_tri = vals_s; T.keys = keys_s; T.n = n;
    const uint32_t b256 = div_up(n, 256u);
    hipLaunchKernelGGL(k_lbvh_morton, dim3(b256), dim3(256), 0, s, sg->d, n, make_float3(blo[0], blo[1], blo[2]), make_float3(binv[0], binv[1], binv[2]), keys, vals);
    void *cub_tmp = scratch(std::max(sort_bytes, scan_bytes));
    if (!cub_tmp) return done(fail(LPT_ERR_HIP, "GPU BVH build: out of device memory"));
    size_t cub_bytes = std::max(sort_bytes, scan_bytes);
    if (hipcub::DeviceRadixSort::SortPairs(cub_tmp, cub_bytes, keys, keys_s, vals, vals_s, (int)n, 0, 30, s) != hipSuccess) return done(fail(LPT_ERR_HIP, "GPU BVH build: sort failed"));
    hipLaunchKernelGGL(k_lbvh_leaf_boxes, dim3(b256), dim3(256), 0, s, sg->d, T);
    LB_TRY(hipMemsetAsync(t_flag, 0, sizeof(uint32_t) * n, s));
    hipLaunchKernelGGL(k_lbvh_tree, dim3(b256), dim3(256), 0, s, T);
    hipLaunchKernelGGL(k_lbvh_fit, dim3(b256), dim3(256), 0, s, T);

    // wide tree, level by level
    SCR(uint4, nodes_big, 4u * (size_t)n)           // <= n-1 wide nodes
    SCR(uint32_t, leaf_big, (size_t)kNodeTris * n)  // their triangle places (16 per node; the final array is cut to the nodes there are)
    SCR(int, items_a, n) SCR(int, items_b, n) SCR(int, kid_ref, 8u * (size_t)n)
    SCR(uint32_t, inner_count, n) SCR(uint32_t, tri_count, n) SCR(uint32_t, inner_off, n) SCR(uint32_t, tri_off, n)
    LB_TRY(hipMemsetAsync(leaf_big, 0xFF, sizeof(uint32_t) * (size_t)kNodeTris * n, s));   // holes: LPT_INVALID_INDEX
    LB_TRY(hipMemsetAsync(items_a, 0, sizeof(int), s));  // level 0 = the binary root
    int *items_cur = items_a, *items_next = items_b;
    uint32_t n_items = 1, level_first = 0, tri_running = 0;
    sg->level_start.clear();
    while (n_items) {
        if (sg->level_start.size() > 64) return done(fail(LPT_ERR_ACCEL_BUILD, "GPU BVH build: tree deeper than 64 levels"));
        LbvhLevel L{items_cur, n_items, kid_ref, inner_count, tri_count};
        hipLaunchKernelGGL(k_lbvh_collapse, dim3(div_up(n_items, 64u)), dim3(64), 0, s, T, L);
        cub_bytes = std::max(sort_bytes, scan_bytes);
        hipcub::DeviceScan::ExclusiveSum(cub_tmp, cub_bytes, inner_count, inner_off, (int)n_items, s);
        cub_bytes = std::max(sort_bytes, scan_bytes);
        hipcub::DeviceScan::ExclusiveSum(cub_tmp, cub_bytes, tri_count, tri_off, (int)n_items, s);
        uint32_t tail[4];
        LB_TRY(hipMemcpyAsync(&tail[0], inner_off + (n_items - 1), 4, hipMemcpyDeviceToHost, s));
        LB_TRY(hipMemcpyAsync(&tail[1], inner_count + (n_items - 1), 4, hipMemcpyDeviceToHost, s));
        LB_TRY(hipMemcpyAsync(&tail[2], tri_off + (n_items - 1), 4, hipMemcpyDeviceToHost, s));
        LB_TRY(hipMemcpyAsync(&tail[3], tri_count + (n_items - 1), 4, hipMemcpyDeviceToHost, s));
        LB_TRY(hipStreamSynchronize(s));
        const uint32_t inner_total = tail[0] + tail[1], tri_total = tail[2] + tail[3];
        if ((size_t)level_first + n_items + inner_total > n) return done(fail(LPT_ERR_ACCEL_BUILD, "GPU BVH build: node budget exceeded"));
        hipLaunchKernelGGL(k_lbvh_emit, dim3(div_up(n_items, 64u)), dim3(64), 0, s, T, L, inner_off, tri_off, level_first, level_first + n_items, tri_running,
                           nodes_big, leaf_big, items_next);
        sg->level_start.push_back(level_first);
        level_first += n_items;
        tri_running += tri_total;
        n_items = inner_total;
        std::swap(items_cur, items_next);
    }
    sg->level_start.push_back(level_first);
    if (tri_running != n) return done(fail(LPT_ERR_ACCEL_BUILD, "GPU BVH build: %u of %u triangles referenced", tri_running, n));
    const uint32_t n_nodes = level_first;
    LB_TRY(hipMalloc(&sg->nodes, sizeof(Node8) * (size_t)n_nodes));
    LB_TRY(hipMemcpyAsync(sg->nodes, nodes_big, sizeof(Node8) * (size_t)n_nodes, hipMemcpyDeviceToDevice, s));
    LB_TRY(hipMalloc(&sg->node_lo, sizeof(float4) * (size_t)n_nodes));
    LB_TRY(hipMalloc(&sg->node_hi, sizeof(float4) * (size_t)n_nodes));
    LB_TRY(hipMalloc(&sg->leaf_prim, sizeof(uint32_t) * (size_t)kNodeTris * n_nodes));
    LB_TRY(hipMemcpyAsync(sg->leaf_prim, leaf_big, sizeof(uint32_t) * (size_t)kNodeTris * n_nodes, hipMemcpyDeviceToDevice, s));
    // Woop maps: prim order (host, double precision) -> their places (holes stay zero)
    LB_TRY(hipMalloc(&sg->woop, sizeof(WoopTri) * (size_t)kNodeTris * n_nodes));
    LB_TRY(hipMemsetAsync(sg->woop, 0, sizeof(WoopTri) * (size_t)kNodeTris * n_nodes, s));
    const float4 *woop_src = dev_woop;
    if (host_woop) {
        SCR(float4, woop_prim, 3u * (size_t)n)
        LB_TRY(hipMemcpyAsync(woop_prim, host_woop, sizeof(WoopTri) * (size_t)n, hipMemcpyHostToDevice, s));
        woop_src = woop_prim;
    }
    hipLaunchKernelGGL(k_lbvh_scatter_woop, dim3(div_up(kNodeTris * n_nodes, 256u)), dim3(256), 0, s, woop_src, (float4 *)sg->woop, (const uint32_t *)sg->leaf_prim, kNodeTris * n_nodes, 0u, n);
    sg->d.nodes = (const DNode8 *)sg->nodes;
    sg->d.leaf_prim = (const uint32_t *)sg->leaf_prim;
    sg->d.woop = (const float4 *)sg->woop;
    for (size_t l = sg->level_start.size() - 1; l-- > 0;) {
        const uint32_t a = sg->level_start[l], b = sg->level_start[l + 1];
        if (b > a) hipLaunchKernelGGL(k_refit_level, dim3(div_up(b - a, 64u)), dim3(64), 0, s, sg->d, (uint4 *)sg->nodes, (float4 *)sg->node_lo, (float4 *)sg->node_hi, a, b);
    }
    LB_TRY(hipGetLastError());
    sg->stats.nodes = n_nodes;
    sg->stats.max_depth = (uint32_t)sg->level_start.size() - 1u;
    const int st = done(LPT_OK);
    sg->stats.build_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return st;
#undef LB_TRY
#undef SCR
}

// The material as the shading records carry it: a material whose albedo and mra textures were paired at upload names the pair.
// Every other texture id is an index into the atlas or LPT_INVALID_INDEX: an id beyond the images means "no texture" (SPEC §9, as
// the oracle reads it) and must not reach the kernel as a value it would take for the paired encoding.  `ok` = false:
// the material samples an image on its own that the atlas does not hold (it was only known as half of a pair at upload time).
static lpt_material device_material(const lpt_scene_gpu *sg, const lpt_material &m, bool *ok = nullptr) {
    lpt_material d = m;
    if (ok) *ok = true;
    const auto it = sg->pair_map.find(std::make_pair(m.albedo_texture, m.mra_texture));
    if (it != sg->pair_map.end()) {
        const DImage &di = sg->pair_descs[it->second];
        d.albedo_texture = kPairedBit | di.offset;          // in 8-byte texels, < 2^30
        d.mra_texture = di.width | (di.height << 16);       // both <= 65535 (checked when the pair is built)
        return d;
    }
    uint32_t *ids[2] = {&d.albedo_texture, &d.mra_texture};
    for (uint32_t *id : ids) {
        if (*id >= sg->image_resident.size()) *id = LPT_INVALID_INDEX;
        else if (!sg->image_resident[*id] && ok) *ok = false;
    }
    return d;
}

// kernel arguments for re-baking instance i on the device (k_bake_instance)
static int make_bake_args(const lpt_scene_gpu *sg, const lpt_scene &scene, size_t i, uint32_t first, uint32_t n, BakeArgs &a) {
    const lpt_instance &now = scene.instances[i];
    const lpt_blas_entry &e = scene.entries[now.blas_index];
    if (e.index_count / 3u != n) return fail(LPT_ERR_INVALID_ARG, "triangle count of instance %zu changed since the upload", i);
    memcpy(a.m, now.model_to_world, sizeof a.m);
    const float *m = a.m;
    const float a00 = m[0], a10 = m[1], a20 = m[2], a01 = m[4], a11 = m[5], a21 = m[6], a02 = m[8], a12 = m[9], a22 = m[10];
    a.c[0] = a11 * a22 - a12 * a21; a.c[1] = a12 * a20 - a10 * a22; a.c[2] = a10 * a21 - a11 * a20;   // bvh.cpp bake_one
    a.c[3] = a02 * a21 - a01 * a22; a.c[4] = a00 * a22 - a02 * a20; a.c[5] = a01 * a20 - a00 * a21;
    a.c[6] = a01 * a12 - a02 * a11; a.c[7] = a02 * a10 - a00 * a12; a.c[8] = a00 * a11 - a01 * a10;
    a.vertex_offset = e.vertex_offset; a.index_offset = e.index_offset; a.first_tri = first; a.n_tris = n;
    bool resident = true;
    const lpt_material dm = device_material(sg, scene.materials[now.material_index < scene.materials.size() ? now.material_index : 0u], &resident);
    if (!resident) return fail(LPT_ERR_INVALID_ARG, "instance %zu now samples a texture on its own that was uploaded only as half of an (albedo, mra) pair: upload the scene again", i);
    memcpy(a.mat, &dm, 32);
    return LPT_OK;
}

extern "C" {

// ============================================================================ Device
int lpt_device_create(int hip_ordinal, lpt_device **out) {
    if (!out) return fail(LPT_ERR_INVALID_ARG, "lpt_device_create: null out");
    // A renderer spreads its work over several HIP streams (wavefront lanes, several renderers in flight).  The ROCm runtime
    // maps streams onto GPU_MAX_HW_QUEUES hardware queues (default 4) and streams that share a queue serialise: a host that
    // keeps more than two frames in flight sets GPU_MAX_HW_QUEUES=8 in its environment BEFORE its first HIP call
    // (INTEGRATION.md §8).  The library does not touch the process environment.
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0)
        return fail(LPT_ERR_HIP, "no HIP device visible (%s); this library has no CPU fallback", e == hipSuccess ? "count = 0" : hipGetErrorString(e));
    if (hip_ordinal < 0 || hip_ordinal >= n) return fail(LPT_ERR_INVALID_ARG, "HIP ordinal %d out of range (%d devices)", hip_ordinal, n);
    HIP_TRY(hipSetDevice(hip_ordinal));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, hip_ordinal));
    lpt_device *d = new (std::nothrow) lpt_device();
    if (!d) return fail(LPT_ERR_INVALID_ARG, "out of host memory");
    d->ordinal = hip_ordinal;
    d->compute_units = prop.multiProcessorCount;
    snprintf(d->name, sizeof d->name, "%s (%s)", prop.name, prop.gcnArchName);
    hipError_t se = hipStreamCreateWithFlags(&d->stream, hipStreamNonBlocking);
    if (se != hipSuccess) { delete d; return fail(LPT_ERR_HIP, "hipStreamCreate failed: %s", hipGetErrorString(se)); }
    *out = d;
    return LPT_OK;
}

int lpt_device_destroy(lpt_device *dev) {
    if (!dev) return LPT_OK;
    hipSetDevice(dev->ordinal);
    hipStreamSynchronize(dev->stream);
    hipStreamDestroy(dev->stream);
    delete dev;
    return LPT_OK;
}

int lpt_device_synchronize(lpt_device *dev) {
    if (!dev) return fail(LPT_ERR_INVALID_ARG, "lpt_device_synchronize: null");
    HIP_TRY(hipStreamSynchronize(dev->stream));
    return LPT_OK;
}

int lpt_device_info(lpt_device *dev, char *name, size_t cap, int *cus) {
    if (!dev) return fail(LPT_ERR_INVALID_ARG, "lpt_device_info: null");
    if (name && cap) snprintf(name, cap, "%s", dev->name);
    if (cus) *cus = dev->compute_units;
    return LPT_OK;
}

int lpt_device_stream(lpt_device *dev, void **stream) {
    if (!dev || !stream) return fail(LPT_ERR_INVALID_ARG, "lpt_device_stream: null");
    *stream = (void *)dev->stream;
    return LPT_OK;
}

// ============================================================================ SceneGPU
int lpt_scene_gpu_destroy(lpt_scene_gpu *sg) {
    if (!sg) return LPT_OK;
    hipSetDevice(sg->dev->ordinal);
    flush_device(sg->dev);
    hipDeviceSynchronize();   // renderers trace on their own streams: frames still in flight read what is freed here
    // a renderer still bound to this scene is detached (a host that drops the scene first must not leave the renderer with a dangling
    // pointer): its next raytrace() is the no-op of a renderer without resources (renderer.rs:403-407) until set_resources is called again
    for (lpt_renderer *r : sg->dev->renderers)
        if (r->sg == sg) { r->sg = nullptr; r->resources_set = false; }
    void *ptrs[] = {sg->nodes, sg->woop, sg->leaf_prim, sg->tri_verts, sg->materials, sg->lights, sg->texels, sg->images, sg->srgb_lut,
                    sg->node_lo, sg->node_hi, sg->obj_verts, sg->obj_indices, sg->bad_flag, sg->arena, sg->pair_texels, sg->pair_images};
    for (void *p : ptrs) if (p) hipFree(p);
    delete sg;
    return LPT_OK;
}

int lpt_scene_upload(lpt_device *dev, const lpt_scene *scene, lpt_scene_gpu **out) { return lpt_scene_upload_ex(dev, scene, LPT_ACCEL_BUILD_HOST_SAH, out); }

int lpt_scene_upload_ex(lpt_device *dev, const lpt_scene *scene, uint32_t flags, lpt_scene_gpu **out) {
    if (!dev || !scene || !out) return fail(LPT_ERR_INVALID_ARG, "lpt_scene_upload: null");
    if ((flags & ~LPT_UPLOAD_NO_TEXTURE_PAIRS) > LPT_ACCEL_BUILD_GPU_LBVH) return fail(LPT_ERR_INVALID_ARG, "lpt_scene_upload_ex: unknown flags %u", flags);
    const bool pair_textures = !(flags & LPT_UPLOAD_NO_TEXTURE_PAIRS);
    flags &= ~LPT_UPLOAD_NO_TEXTURE_PAIRS;
    HIP_TRY(hipSetDevice(dev->ordinal));
    const auto t_upload = std::chrono::steady_clock::now();
    Accel acc;
    bool gpu_build = flags == LPT_ACCEL_BUILD_GPU_LBVH;
    uint32_t n_tris = 0;
    int st = LPT_OK;
    if (gpu_build) {
        // layout only: which run of baked triangles every instance owns (SPEC §2.5 order).  No vertex is touched on the
        // host; the instances are baked on the device below (k_bake_instance)
        acc.inst_first.assign(scene->instances.size(), 0u);
        acc.inst_count.assign(scene->instances.size(), 0u);
        size_t total = 0;
        for (size_t i = 0; i < scene->instances.size(); ++i) {
            const lpt_instance &in = scene->instances[i];
            const uint32_t cnt = in.blas_index < scene->entries.size() ? scene->entries[in.blas_index].index_count / 3u : 0u;
            acc.inst_first[i] = (uint32_t)total;
            acc.inst_count[i] = cnt;
            total += cnt;
        }
        if (total >= (1u << 29)) return fail(LPT_ERR_ACCEL_BUILD, "too many triangles (%zu)", total);
        n_tris = (uint32_t)total;
        if (n_tris < 16u) gpu_build = false;   // tiny scenes: the host builder (a radix tree needs >= 2 leaves)
    }
    if (!gpu_build) {
        st = bake_and_build(*scene, acc);
        if (st != LPT_OK) return st;
        n_tris = (uint32_t)acc.tri_material.size();
    }
    lpt_scene_gpu *sg = new lpt_scene_gpu();
    sg->dev = dev;
    hipStream_t s = dev->stream;
#define UP(field, vec)                                                         \
    if ((st = upload(&sg->field, (vec).data(), (vec).size(), s)) != LPT_OK) {  \
        lpt_scene_gpu_destroy(sg);                                             \
        return st;                                                             \
    }
    static_assert(sizeof(float4) * kTriRec == 128 && sizeof(lpt_vertex) * 3 == 96 && sizeof(lpt_material) == 32, "shading record layout");
    // Paired textures: a material that has both an albedo and an mra texture of one size gets the two interleaved — 8 B per texel
    // (albedo RGBA8, mra RGBA8) in apron tiles of 128 B — so that ONE set of four taps, in ONE cache line, serves both lookups of a shaded hit:
    // 1 line per hit instead of 2 x 1.4 (1.56 with plain 4x4 tiles) (k_shade is bound by HBM traffic, most of it texels).  Lossless: the taps and
    // the filter arithmetic are those of two separate lookups.  An image that some material samples on its own (no partner, a partner
    // of another size, another partner's pair over the budget) also stays in the tiled atlas; one that only ever appears as half of a
    // pair does not (scene.rs:172-184: each image once) — `image_resident`.
    std::vector<DImage> pair_descs;
    std::vector<uint64_t> pair_texels;
    {
        const bool enable = pair_textures;
        const size_t budget = (size_t)1 << 30;   // bytes of paired texels per scene
        for (const lpt_material &m : scene->materials) {
            if (!enable) break;
            const uint32_t a = m.albedo_texture, r = m.mra_texture;
            if (a >= scene->images.size() || r >= scene->images.size()) continue;
            const Image &ia = scene->images[a], &ir = scene->images[r];
            if (ia.width != ir.width || ia.height != ir.height || !ia.width || !ia.height || ia.width > 65535u || ia.height > 65535u) continue;
            if (sg->pair_map.count(std::make_pair(a, r))) continue;
            // apron tiles (kernels.h texture_lookup_pair): a stored 4x4 tile covers a 3x3 block of the image + its right / lower neighbours
            const uint32_t tx = (ia.width + 2u) / 3u, ty = (ia.height + 2u) / 3u;
            if ((pair_texels.size() + (size_t)tx * ty * 16u) * 8u > budget || pair_texels.size() + (size_t)tx * ty * 16u > 0x3FFFFFFFull) continue;
            DImage di;
            di.offset = (uint32_t)pair_texels.size();   // in 8-byte texels
            di.width = ia.width; di.height = ia.height; di.pad = tx;
            const size_t base = pair_texels.size();
            pair_texels.resize(base + (size_t)tx * ty * 16u, 0);
            for (uint32_t tyi = 0; tyi < ty; ++tyi)
                for (uint32_t txi = 0; txi < tx; ++txi)
                    for (uint32_t j = 0; j < 4u; ++j)
                        for (uint32_t i = 0; i < 4u; ++i) {
                            const uint32_t x = (3u * txi + i) % ia.width, y = (3u * tyi + j) % ia.height;   // SPEC §9: repeat wrap
                            uint32_t wa, wr;
                            memcpy(&wa, &ia.rgba8[4u * ((size_t)y * ia.width + x)], 4);
                            memcpy(&wr, &ir.rgba8[4u * ((size_t)y * ir.width + x)], 4);
                            pair_texels[base + ((size_t)tyi * tx + txi) * 16u + j * 4u + i] = (uint64_t)wa | ((uint64_t)wr << 32);
                        }
            sg->pair_map[std::make_pair(a, r)] = (uint32_t)pair_descs.size();
            pair_descs.push_back(di);
            sg->pair_descs.push_back(di);
        }
        // an image no material references is kept: a later material edit (lpt_scene_gpu_update_instances) may start to sample it
        std::vector<uint8_t> referenced(scene->images.size(), 0), alone(scene->images.size(), 0);
        for (const lpt_material &m : scene->materials) {
            const bool paired = sg->pair_map.count(std::make_pair(m.albedo_texture, m.mra_texture)) != 0;
            for (uint32_t id : {m.albedo_texture, m.mra_texture})
                if (id < scene->images.size()) { referenced[id] = 1; if (!paired) alone[id] = 1; }
        }
        sg->image_resident.assign(scene->images.size(), 1);
        for (size_t i = 0; i < scene->images.size(); ++i) sg->image_resident[i] = (!referenced[i] || alone[i]) ? 1 : 0;
    }
    if (!gpu_build) {
        UP(nodes, acc.nodes)
        UP(woop, acc.woop)
        UP(leaf_prim, acc.leaf_prim)
        // shading records (kernels.h DScene::tri_verts): three vertices + the material, 128 B per triangle
        struct TriRec { float4 v[kTriRec]; };
        std::vector<TriRec> recs(acc.tri_material.size());
        for (size_t t = 0; t < recs.size(); ++t) {
            memcpy(recs[t].v, &acc.tri_verts[3 * t], 96);
            const lpt_material dm = device_material(sg, scene->materials[acc.tri_material[t] < scene->materials.size() ? acc.tri_material[t] : 0]);
            memcpy(&recs[t].v[6], &dm, 32);
        }
        UP(tri_verts, recs)
    } else {
        hipError_t e = hipMalloc(&sg->tri_verts, sizeof(float4) * kTriRec * (size_t)n_tris);   // written by k_bake_instance
        if (e != hipSuccess) { lpt_scene_gpu_destroy(sg); return fail(LPT_ERR_HIP, "scene upload failed: %s", hipGetErrorString(e)); }
    }
    UP(materials, scene->materials)
    UP(lights, scene->lights)
    std::vector<DImage> descs;
    std::vector<uint8_t> texels;
    for (size_t ii = 0; ii < scene->images.size(); ++ii) {
        const Image &im = scene->images[ii];
        // texels are stored in 8x4-texel tiles of 128 B (one cache line): the 2x2 bilinear footprint then touches
        // 1.4 lines on average instead of 2 (k_shade is bound by L2-miss traffic); `pad` = tiles per row
        DImage di;
        di.offset = (uint32_t)(texels.size() / 4);
        if (!sg->image_resident[ii]) {   // only ever half of a pair: lives in pair_texels, no shading record carries its id
            di.width = di.height = di.pad = 0u;
            descs.push_back(di);
            continue;
        }
        di.width = im.width; di.height = im.height;
        const uint32_t tx = (im.width + 7u) / 8u, ty = (im.height + 3u) / 4u;
        di.pad = tx;
        descs.push_back(di);
        const size_t base = texels.size();
        texels.resize(base + (size_t)tx * ty * 128u, 0);
        for (uint32_t y = 0; y < im.height; ++y)
            for (uint32_t x = 0; x < im.width; ++x)
                memcpy(&texels[base + 4u * (((size_t)(y >> 2) * tx + (x >> 3)) * 32u + (y & 3u) * 8u + (x & 7u))], &im.rgba8[4u * ((size_t)y * im.width + x)], 4);
    }
    UP(images, descs)
    UP(texels, texels)
    UP(pair_images, pair_descs)
    UP(pair_texels, pair_texels)
    std::vector<float> lut(256);
    for (int i = 0; i < 256; ++i) {
        const double c = (double)i / 255.0;
        lut[i] = (float)(c <= 0.04045 ? c / 12.92 : pow((c + 0.055) / 1.055, 2.4));
    }
    UP(srgb_lut, lut)
    if (!gpu_build) {
        std::vector<float4> boxes(acc.nodes.size(), make_float4(0.f, 0.f, 0.f, 0.f));  // filled by the first refit
        UP(node_lo, boxes)
        UP(node_hi, boxes)
    }
    UP(obj_verts, scene->vertices)
    UP(obj_indices, scene->indices)
    { std::vector<uint32_t> zero(1, 0u); UP(bad_flag, zero) }
#undef UP
    sg->inst_first = acc.inst_first; sg->inst_count = acc.inst_count; sg->level_start = acc.level_start;
    sg->instances = scene->instances;
    sg->n_entries = scene->entries.size(); sg->n_vertices = scene->vertices.size(); sg->n_indices = scene->indices.size();
    hipError_t e = hipStreamSynchronize(s);  // host vectors die at scope exit
    if (e != hipSuccess) { lpt_scene_gpu_destroy(sg); return fail(LPT_ERR_HIP, "scene upload failed: %s", hipGetErrorString(e)); }
    DScene &d = sg->d;
    d.nodes = (const DNode8 *)sg->nodes;
    // a stack entry holds the unvisited siblings below a node at depth 1..max_depth-1 (the root's own group has one
    // member and is never pushed, the deepest nodes have no inner children): max_depth-1 entries are enough
    d.stack_entries = acc.max_depth > 2u ? acc.max_depth - 1u : 1u;
    sg->max_abs = acc.max_abs;      // 0 on the GPU-build path: build_lbvh sets it from its bounds pass
    d.pad_abs = kScenePad * acc.max_abs;
    for (int a = 0; a < 3; ++a) { d.grid_lo[a] = acc.grid_lo[a]; d.grid_step[a] = acc.grid_step[a]; }   // (the GPU-build path: build_lbvh sets its own)
    d.woop = (const float4 *)sg->woop;
    d.leaf_prim = (const uint32_t *)sg->leaf_prim;
    d.tri_verts = (const float4 *)sg->tri_verts;
    d.materials = (const lpt_material *)sg->materials;
    d.lights = (const lpt_light *)sg->lights;
    d.texels = (const uint8_t *)sg->texels;
    d.images = (const DImage *)sg->images;
    d.srgb_lut = (const float *)sg->srgb_lut;
    d.pair_texels = (const uint2 *)sg->pair_texels;
    d.pair_images = (const DImage *)sg->pair_images;
    d.n_pairs = (uint32_t)pair_descs.size();
    d.n_tris = n_tris;
    d.n_materials = (uint32_t)scene->materials.size();
    d.n_lights = (uint32_t)scene->lights.size();
    d.n_images = (uint32_t)scene->images.size();
    sg->stats.triangles = d.n_tris;
    sg->stats.nodes = (uint32_t)acc.nodes.size();
    sg->stats.node_bytes = (uint32_t)sizeof(Node8);
    sg->stats.tri_bytes = (uint32_t)sizeof(WoopTri);
    sg->stats.max_depth = acc.max_depth;
    sg->stats.build_ms = acc.build_ms;
    sg->stats.host_baked_triangles = gpu_build ? 0u : n_tris;
    sg->stats.texture_bytes_resident = (uint64_t)texels.size() + 8u * (uint64_t)pair_texels.size();
    sg->stats.texture_pairs = (uint32_t)pair_descs.size();
    if (gpu_build) {
        // bake every instance on the device (fp32 transform, cofactor normals, binary64 Woop maps: bit-identical to the
        // host bake), then build the tree there
        void *woop_prim = nullptr;
        e = hipMalloc(&woop_prim, sizeof(WoopTri) * (size_t)n_tris);
        if (e != hipSuccess) { lpt_scene_gpu_destroy(sg); return fail(LPT_ERR_HIP, "scene upload failed: %s", hipGetErrorString(e)); }
        for (size_t i = 0; i < scene->instances.size() && st == LPT_OK; ++i) {
            const uint32_t first = sg->inst_first[i], cnt = sg->inst_count[i];
            if (!cnt) continue;
            BakeArgs a;
            st = make_bake_args(sg, *scene, i, first, cnt, a);
            if (st == LPT_OK)
                hipLaunchKernelGGL(k_bake_instance, dim3(div_up(cnt, 256u)), dim3(256), 0, s, a, (const float4 *)sg->obj_verts, (const uint32_t *)sg->obj_indices,
                                   (float4 *)sg->tri_verts, (float4 *)woop_prim, (uint32_t *)sg->bad_flag);
        }
        uint32_t bad = 0;
        if (st == LPT_OK) {
            e = hipMemcpyAsync(&bad, sg->bad_flag, sizeof bad, hipMemcpyDeviceToHost, s);
            if (e == hipSuccess) e = hipStreamSynchronize(s);
            if (e != hipSuccess) st = fail(LPT_ERR_HIP, "scene upload failed: %s", hipGetErrorString(e));
            else if (bad) st = fail(LPT_ERR_ACCEL_BUILD, "non-finite vertex in a baked triangle");
        }
        if (st == LPT_OK) st = build_lbvh(sg, n_tris, nullptr, (const float4 *)woop_prim, s);
        hipFree(woop_prim);
        if (st != LPT_OK) { lpt_scene_gpu_destroy(sg); return st; }
        d.stack_entries = sg->stats.max_depth > 2u ? sg->stats.max_depth - 1u : 1u;
    }
    sg->stats.upload_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t_upload).count();
    *out = sg;
    return LPT_OK;
}

int lpt_scene_gpu_stats(const lpt_scene_gpu *sg, lpt_accel_stats *out) {
    if (!sg || !out) return fail(LPT_ERR_INVALID_ARG, "lpt_scene_gpu_stats: null");
    *out = sg->stats;
    return LPT_OK;
}

// Interactive edit (reference: Instance::set_transform, crates/standalone/src/lib.rs:118-121, followed by a new
// SceneGPU): re-bake the instances whose transform / material changed, upload their triangles and REFIT the wide
// BVH on the GPU (topology kept, all boxes recomputed level by level).  The scene's meshes and instance list
// must be the ones that were uploaded; anything else needs lpt_scene_upload again.
int lpt_scene_gpu_update_instances(lpt_scene_gpu *sg, const lpt_scene *scene, uint32_t *out_rebaked) {
    if (!sg || !scene) return fail(LPT_ERR_INVALID_ARG, "lpt_scene_gpu_update_instances: null");
    if (scene->instances.size() != sg->instances.size() || scene->entries.size() != sg->n_entries ||
        scene->vertices.size() != sg->n_vertices || scene->indices.size() != sg->n_indices)
        return fail(LPT_ERR_INVALID_ARG, "lpt_scene_gpu_update_instances: the scene's meshes / instance list changed since the upload; upload again");
    HIP_TRY(hipSetDevice(sg->dev->ordinal));
    hipStream_t s = sg->dev->stream;
    // renderers trace on their own streams and raytrace() is asynchronous: frames still in flight read the triangles and
    // nodes this call rewrites in place, so wait for every stream of the device first
    { const int fst = flush_device(sg->dev); if (fst != LPT_OK) return fst; }
    HIP_TRY(hipDeviceSynchronize());
    uint32_t changed = 0;
    // the re-baked triangles' Woop maps in prim order; a place-driven scatter then takes them to every place a triangle has in the tree (a split triangle has several)
    void *woop_prim = nullptr;
    const uint32_t n_places = kNodeTris * sg->stats.nodes;
    for (size_t i = 0; i < scene->instances.size(); ++i) {
        const lpt_instance &now = scene->instances[i];
        lpt_instance &was = sg->instances[i];
        if (now.blas_index != was.blas_index)
            return fail(LPT_ERR_INVALID_ARG, "lpt_scene_gpu_update_instances: instance %zu refers to another mesh; upload again", i);
        if (memcmp(&now, &was, sizeof now) == 0) continue;
        const uint32_t first = sg->inst_first[i], n = sg->inst_count[i];
        if (n) {
            // re-bake on the device: the object-space mesh is resident, only the transform travels
            BakeArgs a;
            int bst = make_bake_args(sg, *scene, i, first, n, a);
            if (bst != LPT_OK) { if (woop_prim) hipFree(woop_prim); return bst; }
            if (!woop_prim) HIP_TRY(hipMalloc(&woop_prim, sizeof(WoopTri) * (size_t)std::max(sg->stats.triangles, 1u)));
            hipLaunchKernelGGL(k_bake_instance, dim3(div_up(n, 256u)), dim3(256), 0, s, a, (const float4 *)sg->obj_verts, (const uint32_t *)sg->obj_indices,
                               (float4 *)sg->tri_verts, (float4 *)woop_prim, (uint32_t *)sg->bad_flag);
            hipLaunchKernelGGL(k_lbvh_scatter_woop, dim3(div_up(n_places, 256u)), dim3(256), 0, s, (const float4 *)woop_prim, (float4 *)sg->woop, (const uint32_t *)sg->leaf_prim, n_places, first, n);
        }
        was = now;
        ++changed;
    }
    if (changed) {
        uint32_t bad = 0;
        HIP_TRY(hipMemcpyAsync(&bad, sg->bad_flag, sizeof bad, hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
        if (woop_prim) { hipFree(woop_prim); woop_prim = nullptr; }
        if (bad) {
            HIP_TRY(hipMemsetAsync(sg->bad_flag, 0, sizeof bad, s));
            return fail(LPT_ERR_ACCEL_BUILD, "non-finite vertex in a re-baked instance");
        }
    }
    if (changed && sg->stats.triangles) {
        // the moved triangles' bounds first: the padding's scene-wide part follows the scene's largest coordinate (ANY growth counts — ADVICE r05 —; it never shrinks) and
        // the node origins sit on a grid over the scene — both are inputs of the refit, which then runs once
        float blo[3], bhi[3];
        { const int bst = scene_bounds_and_grid(sg, sg->stats.triangles, s, blo, bhi, true); if (bst != LPT_OK) return bst; }
        for (size_t l = sg->level_start.size() - 1; l-- > 0;) {
            const uint32_t a = sg->level_start[l], b = sg->level_start[l + 1];
            if (b > a) hipLaunchKernelGGL(k_refit_level, dim3(div_up(b - a, 64u)), dim3(64), 0, s, sg->d, (uint4 *)sg->nodes, (float4 *)sg->node_lo, (float4 *)sg->node_hi, a, b);
        }
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipStreamSynchronize(s));
    }
    if (out_rebaked) *out_rebaked = changed;
    return LPT_OK;
}

// Full rebuild on the GPU for scenes whose instances moved far (a refit keeps the old topology and degrades): every
// instance is re-baked on the device and the tree is rebuilt there (build_kernels.h).  Same preconditions as
// lpt_scene_gpu_update_instances; scenes of fewer than 16 triangles are refitted instead.
int lpt_scene_gpu_rebuild(lpt_scene_gpu *sg, const lpt_scene *scene) {
    if (!sg || !scene) return fail(LPT_ERR_INVALID_ARG, "lpt_scene_gpu_rebuild: null");
    if (scene->instances.size() != sg->instances.size() || scene->entries.size() != sg->n_entries ||
        scene->vertices.size() != sg->n_vertices || scene->indices.size() != sg->n_indices)
        return fail(LPT_ERR_INVALID_ARG, "lpt_scene_gpu_rebuild: the scene's meshes / instance list changed since the upload; upload again");
    const uint32_t n = sg->stats.triangles;
    if (n < 16u) return lpt_scene_gpu_update_instances(sg, scene, nullptr);
    HIP_TRY(hipSetDevice(sg->dev->ordinal));
    hipStream_t s = sg->dev->stream;
    { const int fst = flush_device(sg->dev); if (fst != LPT_OK) return fst; }
    HIP_TRY(hipDeviceSynchronize());  // frames in flight on the renderers' streams still read what is rebuilt here
    void *woop_prim = nullptr;
    HIP_TRY(hipMalloc(&woop_prim, sizeof(WoopTri) * (size_t)n));
    for (size_t i = 0; i < scene->instances.size(); ++i) {
        if (scene->instances[i].blas_index != sg->instances[i].blas_index) { hipFree(woop_prim); return fail(LPT_ERR_INVALID_ARG, "lpt_scene_gpu_rebuild: instance %zu refers to another mesh; upload again", i); }
        const uint32_t first = sg->inst_first[i], cnt = sg->inst_count[i];
        if (!cnt) continue;
        BakeArgs a;
        int bst = make_bake_args(sg, *scene, i, first, cnt, a);
        if (bst != LPT_OK) { hipFree(woop_prim); return bst; }
        hipLaunchKernelGGL(k_bake_instance, dim3(div_up(cnt, 256u)), dim3(256), 0, s, a, (const float4 *)sg->obj_verts, (const uint32_t *)sg->obj_indices,
                           (float4 *)sg->tri_verts, (float4 *)woop_prim, (uint32_t *)sg->bad_flag);
    }
    uint32_t bad = 0;
    hipError_t e = hipMemcpyAsync(&bad, sg->bad_flag, sizeof bad, hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (e != hipSuccess || bad) {
        hipMemsetAsync(sg->bad_flag, 0, sizeof bad, s);
        hipFree(woop_prim);
        return e != hipSuccess ? fail(LPT_ERR_HIP, "lpt_scene_gpu_rebuild: %s", hipGetErrorString(e)) : fail(LPT_ERR_ACCEL_BUILD, "non-finite vertex in a re-baked instance");
    }
    // build into NEW buffers and swap them in only on success: a failed build (out of memory, node budget) leaves the
    // scene exactly as it was, so bound renderers and later updates never see freed memory
    void **slots[] = {&sg->nodes, &sg->woop, &sg->leaf_prim, &sg->node_lo, &sg->node_hi};
    void *old[5];
    for (int k = 0; k < 5; ++k) { old[k] = *slots[k]; *slots[k] = nullptr; }
    const DScene old_d = sg->d;
    const lpt_accel_stats old_stats = sg->stats;
    const std::vector<uint32_t> old_levels = sg->level_start;
    int st = build_lbvh(sg, n, nullptr, (const float4 *)woop_prim, s);
    if (st != LPT_OK) {
        for (int k = 0; k < 5; ++k) { if (*slots[k]) hipFree(*slots[k]); *slots[k] = old[k]; }
        sg->d = old_d; sg->stats = old_stats; sg->level_start = old_levels;
        // the shading records were re-baked in place: put the old tree back in step with them (new Woop maps into the
        // old leaf order, boxes refitted; topology kept)
        hipLaunchKernelGGL(k_lbvh_scatter_woop, dim3(div_up(kNodeTris * sg->stats.nodes, 256u)), dim3(256), 0, s, (const float4 *)woop_prim, (float4 *)sg->woop, (const uint32_t *)sg->leaf_prim,
                           kNodeTris * sg->stats.nodes, 0u, n);
        if (sg->stats.triangles) { float blo[3], bhi[3]; scene_bounds_and_grid(sg, n, s, blo, bhi, true); }   // (build_lbvh had set the NEW scene's grid in sg->d before old_d came back)
        if (sg->stats.triangles)
            for (size_t l = sg->level_start.size() - 1; l-- > 0;) {
                const uint32_t a = sg->level_start[l], b = sg->level_start[l + 1];
                if (b > a) hipLaunchKernelGGL(k_refit_level, dim3(div_up(b - a, 64u)), dim3(64), 0, s, sg->d, (uint4 *)sg->nodes, (float4 *)sg->node_lo, (float4 *)sg->node_hi, a, b);
            }
        hipStreamSynchronize(s);
        hipFree(woop_prim);
        sg->instances = scene->instances;  // what is baked now
        return st;
    }
    hipFree(woop_prim);
    for (int k = 0; k < 5; ++k) if (old[k]) hipFree(old[k]);
    sg->d.stack_entries = sg->stats.max_depth > 2u ? sg->stats.max_depth - 1u : 1u;
    sg->instances = scene->instances;
    return LPT_OK;
}

// ============================================================================ ProbeGPU
int lpt_probe_upload(lpt_device *dev, const uint8_t *rgbe8, uint32_t w, uint32_t h, lpt_probe **out) {
    if (!dev || !rgbe8 || !w || !h || !out) return fail(LPT_ERR_INVALID_ARG, "lpt_probe_upload: null or empty");
    HIP_TRY(hipSetDevice(dev->ordinal));
    lpt_probe *p = new lpt_probe();
    p->dev = dev;
    hipError_t e = hipMalloc(&p->rgbe, (size_t)w * h * 4);
    if (e == hipSuccess) e = hipMemcpy(p->rgbe, rgbe8, (size_t)w * h * 4, hipMemcpyHostToDevice);
    if (e != hipSuccess) { if (p->rgbe) hipFree(p->rgbe); delete p; return fail(LPT_ERR_HIP, "probe upload failed: %s", hipGetErrorString(e)); }
    p->d.rgbe = (const uint8_t *)p->rgbe;
    p->d.w = w; p->d.h = h;
    *out = p;
    return LPT_OK;
}

int lpt_probe_destroy(lpt_probe *p) {
    if (!p) return LPT_OK;
    hipSetDevice(p->dev->ordinal);
    flush_device(p->dev);
    hipDeviceSynchronize();   // frames in flight on the renderers' streams may still sample the probe
    for (lpt_renderer *r : p->dev->renderers)   // a renderer still bound to it falls back to the 1x1 default probe (renderer.rs:693-696)
        if (r->probe == p) r->probe = nullptr;
    if (p->rgbe) hipFree(p->rgbe);
    delete p;
    return LPT_OK;
}

// ============================================================================ ray queries
int lpt_trace_closest(lpt_device *dev, const lpt_scene_gpu *sg, const float *origins, const float *dirs, uint32_t n, lpt_hit *out) {
    if (!dev || !sg || (n && (!origins || !dirs || !out))) return fail(LPT_ERR_INVALID_ARG, "lpt_trace_closest: null");
    if (!n) return LPT_OK;
    HIP_TRY(hipSetDevice(dev->ordinal));
    std::vector<float4> o(n), d(n);
    for (uint32_t i = 0; i < n; ++i) {
        o[i] = make_float4(origins[3 * i], origins[3 * i + 1], origins[3 * i + 2], 0.f);
        d[i] = make_float4(dirs[3 * i], dirs[3 * i + 1], dirs[3 * i + 2], -1.f);
    }
    float4 *dO = nullptr, *dD = nullptr, *dH = nullptr;
    int st = LPT_OK;
    hipError_t e = hipMalloc(&dO, sizeof(float4) * n);
    if (e == hipSuccess) e = hipMalloc(&dD, sizeof(float4) * n);
    if (e == hipSuccess) e = hipMalloc(&dH, sizeof(float4) * n);
    if (e == hipSuccess) e = hipMemcpyAsync(dO, o.data(), sizeof(float4) * n, hipMemcpyHostToDevice, dev->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(dD, d.data(), sizeof(float4) * n, hipMemcpyHostToDevice, dev->stream);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_query_closest, dim3(div_up(n, kTraceBlock)), dim3(kTraceBlock), stack_bytes(sg->d), dev->stream, sg->d, dO, dD, dH, n);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(out, dH, sizeof(float4) * n, hipMemcpyDeviceToHost, dev->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(dev->stream);
    if (e != hipSuccess) st = fail(LPT_ERR_HIP, "lpt_trace_closest: %s", hipGetErrorString(e));
    hipFree(dO); hipFree(dD); hipFree(dH);
    return st;
}

int lpt_trace_occluded(lpt_device *dev, const lpt_scene_gpu *sg, const float *origins, const float *dirs, const float *tmax, uint32_t n, uint8_t *out) {
    if (!dev || !sg || (n && (!origins || !dirs || !tmax || !out))) return fail(LPT_ERR_INVALID_ARG, "lpt_trace_occluded: null");
    if (!n) return LPT_OK;
    HIP_TRY(hipSetDevice(dev->ordinal));
    std::vector<float4> o(n), d(n);
    for (uint32_t i = 0; i < n; ++i) {
        o[i] = make_float4(origins[3 * i], origins[3 * i + 1], origins[3 * i + 2], tmax[i]);
        d[i] = make_float4(dirs[3 * i], dirs[3 * i + 1], dirs[3 * i + 2], 0.f);
    }
    float4 *dO = nullptr, *dD = nullptr;
    uint8_t *dR = nullptr;
    int st = LPT_OK;
    hipError_t e = hipMalloc(&dO, sizeof(float4) * n);
    if (e == hipSuccess) e = hipMalloc(&dD, sizeof(float4) * n);
    if (e == hipSuccess) e = hipMalloc(&dR, n);
    if (e == hipSuccess) e = hipMemcpyAsync(dO, o.data(), sizeof(float4) * n, hipMemcpyHostToDevice, dev->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(dD, d.data(), sizeof(float4) * n, hipMemcpyHostToDevice, dev->stream);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_query_occluded, dim3(div_up(n, kTraceBlock)), dim3(kTraceBlock), stack_bytes(sg->d), dev->stream, sg->d, dO, dD, dR, n);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(out, dR, n, hipMemcpyDeviceToHost, dev->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(dev->stream);
    if (e != hipSuccess) st = fail(LPT_ERR_HIP, "lpt_trace_occluded: %s", hipGetErrorString(e));
    hipFree(dO); hipFree(dD); hipFree(dR);
    return st;
}

// ============================================================================ Renderer
static void free_denoiser(lpt_renderer *r) {
    for (int k = 0; k < 2; ++k) {
        if (r->den_gbuf[k]) hipFree(r->den_gbuf[k]);
        if (r->den_rad[k]) hipFree(r->den_rad[k]);
        if (r->den_mom[k]) hipFree(r->den_mom[k]);
        if (r->den_hist[k]) hipFree(r->den_hist[k]);
        r->den_gbuf[k] = nullptr; r->den_rad[k] = nullptr; r->den_mom[k] = nullptr; r->den_hist[k] = nullptr;
    }
    if (r->den_motion) hipFree(r->den_motion);
    if (r->den_temp) hipFree(r->den_temp);
    if (r->den_noisy) hipFree(r->den_noisy);
    if (r->den_nd) hipFree(r->den_nd);
    r->den_motion = nullptr; r->den_temp = nullptr; r->den_noisy = nullptr; r->den_nd = nullptr;
    r->den_inputs_ready = false;
    r->den_cur = 1;
}

static int alloc_denoiser(lpt_renderer *r);
static int ensure_denoiser(lpt_renderer *r) {
    if (r->den_temp) return LPT_OK;
    const int st = alloc_denoiser(r);
    if (st != LPT_OK) free_denoiser(r);   // all or nothing: a later call starts over instead of leaking the part that was allocated
    return st;
}
static int alloc_denoiser(lpt_renderer *r) {
    const size_t n = (size_t)r->w * r->h;
    hipStream_t s = r->stream;
    for (int k = 0; k < 2; ++k) {
        HIP_TRY(hipMalloc(&r->den_gbuf[k], sizeof(uint4) * n));
        HIP_TRY(hipMalloc(&r->den_rad[k], sizeof(float4) * n));
        HIP_TRY(hipMalloc(&r->den_mom[k], sizeof(float2) * n));
        HIP_TRY(hipMalloc(&r->den_hist[k], sizeof(uint32_t) * n));
        HIP_TRY(hipMemsetAsync(r->den_gbuf[k], 0, sizeof(uint4) * n, s));
        HIP_TRY(hipMemsetAsync(r->den_rad[k], 0, sizeof(float4) * n, s));
        HIP_TRY(hipMemsetAsync(r->den_mom[k], 0, sizeof(float2) * n, s));
        HIP_TRY(hipMemsetAsync(r->den_hist[k], 0, sizeof(uint32_t) * n, s));  // history 0 = nothing to reproject
    }
    HIP_TRY(hipMalloc(&r->den_motion, sizeof(float2) * n));
    HIP_TRY(hipMalloc(&r->den_noisy, sizeof(float4) * n));
    HIP_TRY(hipMalloc(&r->den_nd, sizeof(float4) * n));
    HIP_TRY(hipMalloc(&r->den_temp, sizeof(float4) * n));   // the "allocated" marker (ensure_denoiser)
    HIP_TRY(hipMemsetAsync(r->den_noisy, 0, sizeof(float4) * n, s));
    HIP_TRY(hipMemsetAsync(r->den_motion, 0, sizeof(float2) * n, s));
    HIP_TRY(hipStreamSynchronize(s));   // the primary pass of the first frame may run on a lane's stream
    return LPT_OK;
}

static void free_ray_buffers(Wavefront &wf) {
    void *ptrs[] = {wf.q[0].o, wf.q[0].d, wf.q[0].T, wf.q[1].o, wf.q[1].d, wf.q[1].T, wf.sq.o, wf.sq.d, wf.sq.c, wf.hits, wf.Lsum, wf.strag};
    for (void *p : ptrs) if (p) hipFree(p);
    wf.q[0] = Queue{}; wf.q[1] = Queue{}; wf.sq = ShadowQueue{};
    wf.hits = wf.Lsum = nullptr;
    wf.strag = nullptr;
    wf.ray_cap = 0;
}

static void free_frame_buffers(lpt_renderer *r) {
    free_denoiser(r);  // Renderer::resize re-creates the ASVGF resources (renderer.rs:347-355)
    for (int l = 0; l < kMaxLanes; ++l) free_ray_buffers(r->wf[l]);
    if (r->accum) hipFree(r->accum);
    if (r->scratch) hipFree(r->scratch);
    r->accum = r->scratch = nullptr;
    if (r->frame) hipFree(r->frame);
    if (r->xstage) hipFree(r->xstage);
    r->frame = r->xstage = nullptr;
    r->xstage_elems = 0;
    r->xevent_recorded = false;
    r->presented = false;
    r->n_slots = 0;
}

// ---- tile ownership (kernels.h ShardMap / ShardTable).  The V = sum(weights) virtual ranks are dealt to the ranks so that every
// rank's share of any run of tiles follows its weight: virtual rank v goes to the rank that is furthest behind its share
// (ties: the lowest rank).  Unit weights give virtual rank = rank.  Pure host arithmetic, the same on every rank.
static std::vector<uint32_t> deal_virtual_ranks(const std::vector<uint32_t> &weights, uint32_t world) {
    std::vector<uint32_t> owner;
    if (weights.empty()) { owner.resize(world); for (uint32_t q = 0; q < world; ++q) owner[q] = q; return owner; }
    uint32_t V = 0;
    for (uint32_t wq : weights) V += wq;
    std::vector<uint32_t> given(world, 0u);
    for (uint32_t v = 0; v < V; ++v) {
        uint32_t best = 0;
        long long best_deficit = -(1ll << 60);
        for (uint32_t q = 0; q < world; ++q) {
            const long long deficit = (long long)weights[q] * (v + 1u) - (long long)given[q] * V;   // share due - share given, times V
            if (given[q] < weights[q] && deficit > best_deficit) { best = q; best_deficit = deficit; }
        }
        owner.push_back(best);
        given[best]++;
    }
    return owner;
}
static inline uint32_t tiles_of_virtual(uint32_t n_tiles, uint32_t V, uint32_t v) { return n_tiles / V + (v < n_tiles % V ? 1u : 0u); }
static ShardMap make_shard_map(const std::vector<uint32_t> &weights, uint32_t rank, uint32_t world) {
    ShardMap m{};
    const std::vector<uint32_t> owner = deal_virtual_ranks(weights, world);
    m.V = (uint32_t)owner.size();
    for (uint32_t v = 0; v < m.V; ++v)
        if (owner[v] == rank && m.w < kMaxWeight) m.vlist[m.w++] = v;
    return m;
}
static uint32_t owned_tiles(const ShardMap &m, uint32_t n_tiles) {
    uint32_t n = 0;
    for (uint32_t j = 0; j < m.w; ++j) n += tiles_of_virtual(n_tiles, m.V, m.vlist[j]);
    return n;
}
// `offsets` (world + 1 entries, any world size): where every rank's slots start in rank 0's staging area — the host's copy of the rule.
// The device table holds arrays only for a WEIGHTED rule (world <= kMaxWorld, sum of weights <= kMaxVirtual: checked by the callers);
// unit weights are the closed form on both sides (the fixed-size arrays were written for any world).
static ShardTable make_shard_table(const std::vector<uint32_t> &weights, uint32_t world, uint32_t n_tiles, uint32_t area, std::vector<uint32_t> &offsets) {
    ShardTable t{};
    offsets.assign((size_t)world + 1u, 0u);
    if (weights.empty()) {
        t.V = world;
        t.unit_world = world;
        for (uint32_t q = 0; q <= world; ++q) offsets[q] = shard_slot_offset(n_tiles, world, area, q);
        return t;
    }
    const std::vector<uint32_t> owner = deal_virtual_ranks(weights, world);
    t.V = (uint32_t)owner.size();
    std::vector<uint32_t> tiles(world, 0u);
    for (uint32_t v = 0; v < t.V && v < kMaxVirtual; ++v) {
        if (owner[v] >= kMaxWorld) continue;   // unreachable behind the callers' checks; never write past the arrays
        t.owner[v] = (uint8_t)owner[v];
        t.j[v] = (uint8_t)t.w[owner[v]]++;
        tiles[owner[v]] += tiles_of_virtual(n_tiles, t.V, v);
    }
    for (uint32_t q = 0; q < world; ++q) {
        offsets[q + 1] = offsets[q] + tiles[q] * area;
        if (q < kMaxWorld) t.offset[q + 1] = offsets[q + 1];
    }
    return t;
}

static void shard_geometry(const lpt_renderer *r, uint32_t &tiles_x, uint32_t &n_tiles, uint32_t &n_slots) {
    tiles_x = div_up(r->w, r->tile_w);
    const uint32_t tiles_y = div_up(r->h, r->tile_h);
    n_tiles = tiles_x * tiles_y;
    n_slots = owned_tiles(r->map, n_tiles) * r->tile_w * r->tile_h;
}

// per-ray buffers of one lane (queues, hits, shadow queue, per-sample radiance): one element per ray of a wavefront
static int alloc_ray_buffers(Wavefront &wf, size_t rays) {
    free_ray_buffers(wf);
    const size_t n = std::max<size_t>(rays, 64);
    if (n > 0x7FFFFFFFull) return fail(LPT_ERR_INVALID_ARG, "a wavefront of %zu rays exceeds 2^31", n);
    for (int k = 0; k < 2; ++k) {
        HIP_TRY(hipMalloc(&wf.q[k].o, sizeof(float4) * n));
        HIP_TRY(hipMalloc(&wf.q[k].d, sizeof(float4) * n));
        HIP_TRY(hipMalloc(&wf.q[k].T, sizeof(float4) * n));
    }
    HIP_TRY(hipMalloc(&wf.sq.o, sizeof(float4) * n));
    HIP_TRY(hipMalloc(&wf.sq.d, sizeof(float4) * n));
    HIP_TRY(hipMalloc(&wf.sq.c, sizeof(float4) * n));
    HIP_TRY(hipMalloc(&wf.hits, sizeof(float4) * n));
    HIP_TRY(hipMalloc(&wf.Lsum, sizeof(float4) * n));
    HIP_TRY(hipMalloc(&wf.strag, sizeof(uint32_t) * 2 * n));   // a launch carries at most n closest-hit + n shadow rays
    wf.ray_cap = n;
    return LPT_OK;
}

// stream, events and counters of a lane (created on first use)
static int ensure_lane(lpt_renderer *r, int l) {
    Wavefront &wf = r->wf[l];
    if (!wf.stream) HIP_TRY(hipStreamCreateWithFlags(&wf.stream, hipStreamNonBlocking));
    if (!wf.done) HIP_TRY(hipEventCreateWithFlags(&wf.done, hipEventDisableTiming));
    if (!wf.consumed) HIP_TRY(hipEventCreateWithFlags(&wf.consumed, hipEventDisableTiming));
    if (!wf.ctr) HIP_TRY(hipMalloc(&wf.ctr, sizeof(FrameCounters)));
    return LPT_OK;
}

static int alloc_frame_buffers(lpt_renderer *r) {
    HIP_TRY(hipSetDevice(r->dev->ordinal));
    HIP_TRY(hipStreamSynchronize(r->stream));
    free_frame_buffers(r);
    if (!r->w || !r->h) return LPT_OK;
    uint32_t tiles_x, n_tiles, n_slots;
    shard_geometry(r, tiles_x, n_tiles, n_slots);
    const size_t px = (size_t)r->w * r->h;
    r->n_slots = n_slots;   // the lanes' ray buffers are allocated by the first raytrace() that uses them
    if (r->rank == 0u) {   // a possible root of an exchange (also of a one-rank communicator): the whole ownership rule, for the unpack kernels
        r->h_table = make_shard_table(r->weights, r->world, n_tiles, r->tile_w * r->tile_h, r->h_offset);
        if (!r->d_table) HIP_TRY(hipMalloc(&r->d_table, sizeof(ShardTable)));
        HIP_TRY(hipMemcpy(r->d_table, &r->h_table, sizeof(ShardTable), hipMemcpyHostToDevice));
    }
    HIP_TRY(hipMalloc(&r->accum, sizeof(float4) * px));
    HIP_TRY(hipMalloc(&r->scratch, sizeof(float4) * px));
    HIP_TRY(hipMemsetAsync(r->accum, 0, sizeof(float4) * px, r->stream));
    return LPT_OK;
}

int lpt_renderer_create(lpt_device *dev, uint32_t width, uint32_t height, lpt_renderer **out) {
    if (!dev || !out) return fail(LPT_ERR_INVALID_ARG, "lpt_renderer_create: null");
    HIP_TRY(hipSetDevice(dev->ordinal));
    lpt_renderer *r = new lpt_renderer();
    r->dev = dev;
    {
        hipError_t se = hipStreamCreateWithFlags(&r->stream, hipStreamNonBlocking);
        if (se != hipSuccess) { delete r; return fail(LPT_ERR_HIP, "hipStreamCreate failed: %s", hipGetErrorString(se)); }
    }
    r->prev_cam.origin = mk3(0.f, 0.f, 0.f); r->prev_cam.right = mk3(1.f, 0.f, 0.f); r->prev_cam.up = mk3(0.f, 1.f, 0.f);
    r->prev_cam.fwd = mk3(0.f, 0.f, 1.f); r->prev_cam.ax = r->prev_cam.ay = 1.0f;   // Mat4::IDENTITY (renderer.rs:319)
    r->req_w = width; r->req_h = height;
    // get_downsampled_size (renderer.rs:18-22)
    r->w = (uint32_t)((float)width * r->downsample);
    r->h = (uint32_t)((float)height * r->downsample);
    hipError_t e = hipMalloc(&r->totals, sizeof(Totals));
    if (e == hipSuccess) e = hipMemset(r->totals, 0, sizeof(Totals));
    if (e == hipSuccess) e = hipHostMalloc((void **)&r->err_host, sizeof(uint32_t), hipHostMallocMapped);
    if (e == hipSuccess) { *r->err_host = 0u; e = hipHostGetDevicePointer((void **)&r->err_dev, r->err_host, 0); }
    if (e == hipSuccess) e = hipMalloc(&r->default_probe, 4);
    if (e == hipSuccess) e = hipMemset(r->default_probe, 0, 4);  // 1x1 zero texel: black environment (device.rs:13-26)
    if (e == hipSuccess) e = hipMalloc(&r->srgb_thr, 256 * sizeof(float));
    if (e == hipSuccess) {
        float thr[256];
        thr[0] = 0.0f;
        for (int i = 1; i < 256; ++i) {   // inverse OETF at (i - 0.5) / 255 in binary64, rounded once
            const double v = ((double)i - 0.5) / 255.0;
            thr[i] = (float)(v <= 0.04045 ? v / 12.92 : pow((v + 0.055) / 1.055, 2.4));
        }
        e = hipMemcpy(r->srgb_thr, thr, sizeof thr, hipMemcpyHostToDevice);
    }
    if (e != hipSuccess) {
        const int st = fail(LPT_ERR_HIP, "renderer allocation failed: %s", hipGetErrorString(e));
        lpt_renderer_destroy(r);   // frees whatever was allocated (every member starts out null)
        return st;
    }
    int st = alloc_frame_buffers(r);
    if (st != LPT_OK) { lpt_renderer_destroy(r); return st; }
    dev->renderers.push_back(r);
    *out = r;
    return LPT_OK;
}

int lpt_renderer_destroy(lpt_renderer *r) {
    if (!r) return LPT_OK;
    hipSetDevice(r->dev->ordinal);
    r->pend.n = 0;   // recorded but never submitted: nobody can read the result any more
    forget_deferred_exchange(r);   // an exchange enqueued inside a still open lpt_comm_group bracket must not outlive the renderer
    {
        auto &v = r->dev->renderers;
        v.erase(std::remove(v.begin(), v.end(), r), v.end());
    }
    hipStreamSynchronize(r->stream);
    for (int l = 0; l < kMaxLanes; ++l)   // a wavefront whose second half was never enqueued (a failed submission) is not behind r->stream
        if (r->wf[l].stream) hipStreamSynchronize(r->wf[l].stream);
    free_frame_buffers(r);
    for (int l = 0; l < kMaxLanes; ++l) {
        Wavefront &wf = r->wf[l];
        if (wf.stream) hipStreamDestroy(wf.stream);
        if (wf.done) hipEventDestroy(wf.done);
        if (wf.consumed) hipEventDestroy(wf.consumed);
        if (wf.ctr) hipFree(wf.ctr);
    }
    if (r->totals) hipFree(r->totals);
    if (r->err_host) hipHostFree(r->err_host);
    if (r->occ_table) hipFree(r->occ_table);
    if (r->d_table) hipFree(r->d_table);
    if (r->default_probe) hipFree(r->default_probe);
    if (r->srgb_thr) hipFree(r->srgb_thr);
    if (r->xevent) hipEventDestroy(r->xevent);
    if (r->stream) hipStreamDestroy(r->stream);
    if (r->noise) hipFree(r->noise);
    if (r->ev_start) {
        for (int i = 0; i < lpt_renderer::kRing * lpt_renderer::kMaxEvents; ++i) {
            if (r->ev_start[i]) hipEventDestroy(r->ev_start[i]);
            if (r->ev_stop[i]) hipEventDestroy(r->ev_stop[i]);
        }
        delete[] r->ev_start;
        delete[] r->ev_stop;
    }
    delete r;
    return LPT_OK;
}

int lpt_renderer_set_downsample(lpt_renderer *r, float factor) {
    if (!r || !(factor > 0.f) || factor > 16.f) return fail(LPT_ERR_INVALID_ARG, "lpt_renderer_set_downsample: factor must be in (0,16]");
    FLUSH_OR_RETURN(r);
    r->downsample = factor;
    return LPT_OK;
}

int lpt_renderer_set_resources(lpt_renderer *r, const lpt_scene_gpu *sg, const lpt_probe *probe) {
    if (!r || !sg) return fail(LPT_ERR_INVALID_ARG, "lpt_renderer_set_resources: null");
    if (sg->dev != r->dev || (probe && probe->dev != r->dev)) return fail(LPT_ERR_INVALID_ARG, "resources belong to another device");
    FLUSH_OR_RETURN(r);
    r->sg = sg;
    r->probe = probe;
    r->resources_set = true;
    r->frame_count = 1;  // renderer.rs:724
    return LPT_OK;
}

int lpt_renderer_resize(lpt_renderer *r, const lpt_scene_gpu *sg, const lpt_probe *probe, uint32_t width, uint32_t height) {
    if (!r || !sg) return fail(LPT_ERR_INVALID_ARG, "lpt_renderer_resize: null");
    FLUSH_OR_RETURN(r);
    const uint32_t nw = (uint32_t)((float)width * r->downsample), nh = (uint32_t)((float)height * r->downsample);
    if (nw > 8192u || nh > 8192u) return fail(LPT_ERR_INVALID_ARG, "lpt_renderer_resize: the path-traced size is limited to 8192 x 8192 (got %u x %u)", nw, nh);
    r->req_w = width; r->req_h = height;
    r->w = nw;
    r->h = nh;
    int st = alloc_frame_buffers(r);
    if (st != LPT_OK) return st;
    return lpt_renderer_set_resources(r, sg, probe);
}

int lpt_renderer_get_size(const lpt_renderer *r, uint32_t *w, uint32_t *h) {
    if (!r) return fail(LPT_ERR_INVALID_ARG, "lpt_renderer_get_size: null");
    if (w) *w = r->w;
    if (h) *h = r->h;
    return LPT_OK;
}

// largest per-pixel record: a queued ray (origin+slot, dir+pdf, throughput) = 48 B
uint32_t lpt_max_per_pixel_bytes(void) { return 48u; }

int lpt_renderer_set_max_bounces(lpt_renderer *r, uint32_t b) {
    if (!r || b == 0 || b > (uint32_t)kMaxBounces) return fail(LPT_ERR_INVALID_ARG, "max_bounces must be in [1,%d]", kMaxBounces);
    FLUSH_OR_RETURN(r);
    r->max_bounces = b;
    return LPT_OK;
}
int lpt_renderer_set_seed(lpt_renderer *r, uint32_t s) {
    if (!r) return fail(LPT_ERR_INVALID_ARG, "lpt_renderer_set_seed: null");
    FLUSH_OR_RETURN(r);
    r->user_seed = s;
    return LPT_OK;
}
int lpt_renderer_set_vfov(lpt_renderer *r, float radians) {
    if (!r || !(radians > 0.f) || !(radians < 3.14159f)) return fail(LPT_ERR_INVALID_ARG, "vfov must be in (0, pi)");
    FLUSH_OR_RETURN(r);
    r->vfov = radians;
    return LPT_OK;
}
int lpt_renderer_set_shard_weighted(lpt_renderer *r, uint32_t rank, uint32_t world, uint32_t tile_w, uint32_t tile_h, const uint32_t *weights) {
    if (!r || world == 0 || rank >= world || tile_w == 0 || tile_h == 0 || (tile_w * tile_h) % 64u != 0u)
        return fail(LPT_ERR_INVALID_ARG, "bad shard (rank %u of %u, tile %ux%u; tile area must be a multiple of 64)", rank, world, tile_w, tile_h);
    std::vector<uint32_t> wv;
    if (weights) {
        if (world > kMaxWorld) return fail(LPT_ERR_INVALID_ARG, "weighted shards: at most %u ranks", kMaxWorld);
        uint32_t sum = 0;
        bool unit = true;
        for (uint32_t q = 0; q < world; ++q) {
            if (weights[q] > kMaxWeight) return fail(LPT_ERR_INVALID_ARG, "weighted shards: weight %u of rank %u exceeds %u", weights[q], q, kMaxWeight);
            sum += weights[q];
            unit = unit && weights[q] == 1u;
        }
        if (sum == 0u || sum > kMaxVirtual) return fail(LPT_ERR_INVALID_ARG, "weighted shards: the weights must sum to 1..%u (got %u)", kMaxVirtual, sum);
        if (!unit) wv.assign(weights, weights + world);
    }
    FLUSH_OR_RETURN(r);
    r->rank = rank; r->world = world; r->tile_w = tile_w; r->tile_h = tile_h;
    r->weights = wv;
    r->map = make_shard_map(r->weights, rank, world);
    r->frame_count = 1;
    return alloc_frame_buffers(r);
}
int lpt_renderer_set_shard(lpt_renderer *r, uint32_t rank, uint32_t world, uint32_t tile_w, uint32_t tile_h) {
    return lpt_renderer_set_shard_weighted(r, rank, world, tile_w, tile_h, nullptr);
}
int lpt_renderer_reset_accumulation(lpt_renderer *r) {
    if (!r) return fail(LPT_ERR_INVALID_ARG, "lpt_renderer_reset_accumulation: null");
    r->frame_count = 1;       // renderer.rs:610
    r->accumulate = false;    // :611 ; the seed is deliberately not reset (:613-615)
    return LPT_OK;
}
int lpt_renderer_set_accumulate(lpt_renderer *r, int a) {
    if (!r) return fail(LPT_ERR_INVALID_ARG, "lpt_renderer_set_accumulate: null");
    r->accumulate = a != 0;
    return LPT_OK;
}
int lpt_renderer_get_accumulate(const lpt_renderer *r, int *a) {
    if (!r || !a) return fail(LPT_ERR_INVALID_ARG, "lpt_renderer_get_accumulate: null");
    *a = r->accumulate ? 1 : 0;
    return LPT_OK;
}
int lpt_renderer_get_frame_state(const lpt_renderer *r, uint32_t *fc, uint32_t *seed) {
    if (!r) return fail(LPT_ERR_INVALID_ARG, "lpt_renderer_get_frame_state: null");
    if (fc) *fc = r->frame_count;
    if (seed) *seed = r->seed;
    return LPT_OK;
}
int lpt_renderer_upload_noise(lpt_renderer *r, const uint8_t *rgba8, uint32_t w, uint32_t h, uint32_t row_bytes) {
    if (!r || !rgba8 || !w || !h || row_bytes < w * 4u) return fail(LPT_ERR_INVALID_ARG, "lpt_renderer_upload_noise: bad arguments");
    FLUSH_OR_RETURN(r);
    HIP_TRY(hipSetDevice(r->dev->ordinal));
    HIP_TRY(hipStreamSynchronize(r->stream));
    if (r->noise) { hipFree(r->noise); r->noise = nullptr; }
    HIP_TRY(hipMalloc(&r->noise, (size_t)w * h * 4));
    HIP_TRY(hipMemcpy2D(r->noise, (size_t)w * 4, rgba8, row_bytes, (size_t)w * 4, h, hipMemcpyHostToDevice));
    r->noise_w = w; r->noise_h = h;
    return LPT_OK;
}
int lpt_renderer_use_noise(lpt_renderer *r, int flag) {
    if (!r) return fail(LPT_ERR_INVALID_ARG, "lpt_renderer_use_noise: null");
    FLUSH_OR_RETURN(r);
    r->use_noise = flag != 0;
    return LPT_OK;
}
int lpt_renderer_set_blit_mode(lpt_renderer *r, int mode) {
    if (!r || mode < LPT_BLIT_PATHTRACE || mode > LPT_BLIT_MOTION) return fail(LPT_ERR_INVALID_ARG, "lpt_renderer_set_blit_mode: bad mode %d", mode);
    FLUSH_OR_RETURN(r);
    r->mode = mode;
    return LPT_OK;
}
int lpt_renderer_set_lanes(lpt_renderer *r, int lanes) {
    if (!r || lanes < 1 || lanes > kMaxLanes) return fail(LPT_ERR_INVALID_ARG, "lpt_renderer_set_lanes: 1..%d", kMaxLanes);
    FLUSH_OR_RETURN(r);
    HIP_TRY(hipSetDevice(r->dev->ordinal));
    HIP_TRY(hipStreamSynchronize(r->stream));   // behind its waits: every lane's work
    for (int l = 0; l < kMaxLanes; ++l) {
        if (r->wf[l].stream) HIP_TRY(hipStreamSynchronize(r->wf[l].stream));
        r->wf[l].consumed_recorded = false;
        if (l >= lanes) free_ray_buffers(r->wf[l]);
    }
    r->n_lanes = lanes;
    r->lane_rr = 0;
    return LPT_OK;
}
int lpt_renderer_set_sort_queues(lpt_renderer *r, int flag) {
    if (!r) return fail(LPT_ERR_INVALID_ARG, "lpt_renderer_set_sort_queues: null");
    FLUSH_OR_RETURN(r);
    // 1: next-bounce queue, 2: shadow queue; any other non-zero value: both queues
    r->sort_queues = flag ? ((flag & 3) ? (flag & 3) : 3) : 0;
    return LPT_OK;
}
// Launch tuning that experiments and the variant tests switch (the reference has no counterpart: SURVEY §5 "Config / flags: no").
// Every value gives the same frame bit for bit; only which kernels run, and how large their grids are, changes.
int lpt_renderer_set_option(lpt_renderer *r, int option, uint64_t value) {
    if (!r) return fail(LPT_ERR_INVALID_ARG, "lpt_renderer_set_option: null");
    FLUSH_OR_RETURN(r);
    switch (option) {
    case LPT_OPT_PACKET_PRIMARY: if (value > 2u) return fail(LPT_ERR_INVALID_ARG, "LPT_OPT_PACKET_PRIMARY: 0 (never), 1 (always), 2 (by pixel footprint)"); r->packet_primary = (uint32_t)value; break;
    case LPT_OPT_EXPERIMENT(LPT_EXP_PIPE_RAYS): r->pipe_rays = (uint32_t)std::min<uint64_t>(value, 0x7FFFFFFFu); break;
    case LPT_OPT_WAVEFRONT_RAYS: r->wavefront_rays = std::max<uint64_t>(value, 64u); break;
    case LPT_OPT_EXPERIMENT(LPT_EXP_REFILL): if (value > 63u) return fail(LPT_ERR_INVALID_ARG, "LPT_EXP_REFILL: 0..63"); r->refill = (int)value; break;
    case LPT_OPT_EXPERIMENT(LPT_EXP_TRACE_WAVES_PER_CU): if (value > 32u) return fail(LPT_ERR_INVALID_ARG, "LPT_EXP_TRACE_WAVES_PER_CU: 0 (auto) or 1..32"); r->trace_waves_per_cu = (uint32_t)value; break;
    case LPT_OPT_EXPERIMENT(LPT_EXP_SHADE_BLOCKS_PER_CU): if (value > 64u) return fail(LPT_ERR_INVALID_ARG, "LPT_EXP_SHADE_BLOCKS_PER_CU: 0 (by the submission) or 1..64"); r->shade_blocks_per_cu = (uint32_t)value; break;
    case LPT_OPT_PATH_RAYS: r->path_rays = (uint32_t)std::min<uint64_t>(value, 0x7FFFFFFFu); break;
    case LPT_OPT_EXPERIMENT(LPT_EXP_PATH_WAVES_PER_CU): if (value < 1u || value > 32u) return fail(LPT_ERR_INVALID_ARG, "LPT_EXP_PATH_WAVES_PER_CU: 1..32"); r->path_waves_per_cu = (uint32_t)value; break;
    case LPT_OPT_EXPERIMENT(LPT_EXP_PATH_REFILL): if (value > 63u) return fail(LPT_ERR_INVALID_ARG, "LPT_EXP_PATH_REFILL: 0..63"); r->path_refill = (int)value; break;
    case LPT_OPT_EXPERIMENT(LPT_EXP_OCC_CELL_MILLI): r->occ_cell = (float)std::min<uint64_t>(value, 1000000u) * 1.0e-3f; break;
    case LPT_OPT_EXPERIMENT(LPT_EXP_STEP_BUDGET): r->step_budget = (uint32_t)std::min<uint64_t>(value, 1u << 20); break;
    case LPT_OPT_EXPERIMENT(LPT_EXP_BUDGET_RAYS): r->budget_rays = (uint32_t)std::min<uint64_t>(value, 0x7FFFFFFFu); break;
    case LPT_OPT_EXPERIMENT(LPT_EXP_PACKET_QUADS): r->packet_quads = value != 0; break;
    case LPT_OPT_EXPERIMENT(LPT_EXP_SPLIT_RAYS): r->split_rays = value; break;
    case LPT_OPT_EXPERIMENT(LPT_EXP_BUDGET_SPLIT): r->budget_split = value != 0; break;
    case LPT_OPT_TAIL_LANES: r->tail_lanes = (uint32_t)std::min<uint64_t>(value, kTailMax); break;
    case LPT_OPT_COOP_RAYS: r->coop_rays = (uint32_t)std::min<uint64_t>(value, 0x7FFFFFFFu); break;
    default: return fail(LPT_ERR_INVALID_ARG, "lpt_renderer_set_option: unknown option %d", option);
    }
    return LPT_OK;
}
int lpt_renderer_get_option(const lpt_renderer *r, int option, uint64_t *value) {
    if (!r || !value) return fail(LPT_ERR_INVALID_ARG, "lpt_renderer_get_option: null");
    switch (option) {
    case LPT_OPT_PACKET_PRIMARY: *value = r->packet_primary; break;
    case LPT_OPT_EXPERIMENT(LPT_EXP_PIPE_RAYS): *value = r->pipe_rays; break;
    case LPT_OPT_WAVEFRONT_RAYS: *value = r->wavefront_rays; break;
    case LPT_OPT_EXPERIMENT(LPT_EXP_REFILL): *value = (uint64_t)r->refill; break;
    case LPT_OPT_EXPERIMENT(LPT_EXP_TRACE_WAVES_PER_CU): *value = r->trace_waves_per_cu; break;
    case LPT_OPT_EXPERIMENT(LPT_EXP_SHADE_BLOCKS_PER_CU): *value = r->shade_blocks_per_cu; break;
    case LPT_OPT_PATH_RAYS: *value = r->path_rays; break;
    case LPT_OPT_EXPERIMENT(LPT_EXP_PATH_WAVES_PER_CU): *value = r->path_waves_per_cu; break;
    case LPT_OPT_EXPERIMENT(LPT_EXP_PATH_REFILL): *value = (uint64_t)r->path_refill; break;
    case LPT_OPT_EXPERIMENT(LPT_EXP_OCC_CELL_MILLI): *value = (uint64_t)(r->occ_cell * 1000.0f + 0.5f); break;
    case LPT_OPT_EXPERIMENT(LPT_EXP_STEP_BUDGET): *value = r->step_budget; break;
    case LPT_OPT_EXPERIMENT(LPT_EXP_BUDGET_RAYS): *value = r->budget_rays; break;
    case LPT_OPT_EXPERIMENT(LPT_EXP_PACKET_QUADS): *value = r->packet_quads; break;
    case LPT_OPT_EXPERIMENT(LPT_EXP_SPLIT_RAYS): *value = r->split_rays; break;
    case LPT_OPT_EXPERIMENT(LPT_EXP_BUDGET_SPLIT): *value = r->budget_split; break;
    case LPT_OPT_TAIL_LANES: *value = r->tail_lanes; break;
    case LPT_OPT_COOP_RAYS: *value = r->coop_rays; break;
    default: return fail(LPT_ERR_INVALID_ARG, "lpt_renderer_get_option: unknown option %d", option);
    }
    return LPT_OK;
}
int lpt_renderer_enable_stats(lpt_renderer *r, int flag) {
    if (!r) return fail(LPT_ERR_INVALID_ARG, "lpt_renderer_enable_stats: null");
    FLUSH_OR_RETURN(r);
    r->stats = flag != 0;
    if (flag) {   // the occluder-cache probe's table (kernels.h OccProbe): empty at the start of every stats session
        HIP_TRY(hipSetDevice(r->dev->ordinal));
        if (!r->occ_table) HIP_TRY(hipMalloc(&r->occ_table, sizeof(uint32_t) * kOccEntries));
        HIP_TRY(hipMemsetAsync(r->occ_table, 0, sizeof(uint32_t) * kOccEntries, r->stream));
        HIP_TRY(hipStreamSynchronize(r->stream));
    }
    return LPT_OK;
}
int lpt_renderer_enable_timings(lpt_renderer *r, int flag) {
    if (!r) return fail(LPT_ERR_INVALID_ARG, "lpt_renderer_enable_timings: null");
    FLUSH_OR_RETURN(r);
    HIP_TRY(hipSetDevice(r->dev->ordinal));
    const int total = lpt_renderer::kRing * lpt_renderer::kMaxEvents;
    if (flag && !r->ev_start) {
        r->ev_start = new hipEvent_t[total]();   // null until created: a failure half way leaves nothing undefined for destroy
        r->ev_stop = new hipEvent_t[total]();
    }
    if (flag) {
        for (int i = 0; i < total; ++i) {
            if (!r->ev_start[i]) HIP_TRY(hipEventCreate(&r->ev_start[i]));
            if (!r->ev_stop[i]) HIP_TRY(hipEventCreate(&r->ev_stop[i]));
        }
    }
    if (flag) {  // (re)start accumulating
        HIP_TRY(hipStreamSynchronize(r->stream));
        for (int k = 0; k < lpt_renderer::kRing; ++k) r->ev_count[k] = 0;
        for (int i = 0; i < ST_COUNT; ++i) { r->stage_ms[i] = 0.0; r->stage_launches[i] = 0; }
    }
    r->timings = flag != 0;
    return LPT_OK;
}

// fold one ring slot's finished events into the per-stage totals (waits for them if needed)
static void harvest_slot(lpt_renderer *r, int slot) {
    const int n = r->ev_count[slot];
    for (int i = 0; i < n; ++i) {
        const int e = slot * lpt_renderer::kMaxEvents + i;
        float ms = 0.f;
        hipEventSynchronize(r->ev_stop[e]);
        if (hipEventElapsedTime(&ms, r->ev_start[e], r->ev_stop[e]) == hipSuccess) {
            r->stage_ms[r->ev_stage[slot][i]] += ms;
            r->stage_launches[r->ev_stage[slot][i]]++;
        }
    }
    r->ev_count[slot] = 0;
}
static inline int cur_slot(const lpt_renderer *r) { return (int)(r->ring_pos % lpt_renderer::kRing); }
static inline void stage_begin(lpt_renderer *r, int stage, hipStream_t stream) {
    if (!r->timings) return;
    const int slot = cur_slot(r);
    if (r->ev_count[slot] >= lpt_renderer::kMaxEvents) return;
    r->ev_stage[slot][r->ev_count[slot]] = stage;
    hipEventRecord(r->ev_start[slot * lpt_renderer::kMaxEvents + r->ev_count[slot]], stream);
}
static inline void stage_end(lpt_renderer *r, hipStream_t stream) {
    if (!r->timings) return;
    const int slot = cur_slot(r);
    if (r->ev_count[slot] >= lpt_renderer::kMaxEvents) return;
    hipEventRecord(r->ev_stop[slot * lpt_renderer::kMaxEvents + r->ev_count[slot]], stream);
    r->ev_count[slot]++;
}

}  // extern "C"

// asvgf.render (renderer.rs:513-518, asvgf.rs:250-291) / asvgf.temporal_pass (:519-522) over the whole frame, from the
// per-pixel inputs (noisy radiance, G-buffer, motion) of the current frame
static void launch_filter(lpt_renderer *r, hipStream_t s) {
    if (r->mode != LPT_BLIT_DENOISED && r->mode != LPT_BLIT_TEMPORAL) return;  // debug views read the inputs directly (:539)
    const int cur = r->den_cur, prv = 1 - r->den_cur;
    const uint32_t npx = r->w * r->h;
    const uint32_t px_blocks = div_up(npx, kBlock);
    const uint32_t stream_blocks = std::min<uint32_t>(px_blocks, (uint32_t)r->dev->compute_units * 8u);
    hipLaunchKernelGGL(k_temporal, dim3(stream_blocks), dim3(kBlock), 0, s, (int)r->w, (int)r->h, r->den_noisy, r->den_gbuf[cur], r->den_gbuf[prv], r->den_motion,
                       r->den_rad[prv], r->den_mom[prv], r->den_hist[prv], r->den_rad[cur], r->den_mom[cur], r->den_hist[cur]);
    const float4 *result = r->den_rad[cur];
    if (r->mode == LPT_BLIT_DENOISED) {
        hipMemcpyAsync(r->den_temp, r->den_rad[cur], sizeof(float4) * npx, hipMemcpyDeviceToDevice, s);  // copy_texture_to_texture
        // even number of a-trous calls: main <-> radiance_temp, result ends in radiance_temp (asvgf.rs:286-287)
        hipLaunchKernelGGL(k_decode_gbuf, dim3(px_blocks), dim3(kBlock), 0, s, r->den_gbuf[cur], r->den_nd, npx);
        hipLaunchKernelGGL(k_atrous, dim3(px_blocks), dim3(kBlock), 0, s, r->den_nd, r->den_temp, r->accum, (int)r->w, (int)r->h, 1);
        hipLaunchKernelGGL(k_atrous, dim3(px_blocks), dim3(kBlock), 0, s, r->den_nd, r->accum, r->den_temp, (int)r->w, (int)r->h, 2);
        hipLaunchKernelGGL(k_atrous, dim3(px_blocks), dim3(kBlock), 0, s, r->den_nd, r->den_temp, r->accum, (int)r->w, (int)r->h, 4);
        hipLaunchKernelGGL(k_atrous, dim3(px_blocks), dim3(kBlock), 0, s, r->den_nd, r->accum, r->den_temp, (int)r->w, (int)r->h, 8);
        result = r->den_temp;
    }
    hipLaunchKernelGGL(k_composite, dim3(px_blocks), dim3(kBlock), 0, s, r->den_gbuf[cur], result, r->accum, npx);
}

extern "C" {

// A wavefront between its two halves (flush_pending): what the second half needs to know about the first
struct Ticket {
    FrameParams p;
    int lane = 0;
    bool split = false, denoise = false, packet = false;
    CamBasis cur{};
};

// First half of a wavefront: the launches of `n_samples` recorded raytrace() calls over the rank's pixel slots
// [slot0, slot0 + piece_slots) — ray generation, traversal, shading — on the wavefront's lane, from the protocol state the first
// of the calls saw (frame_count0, seed0, acc0 = its accumulate flag; the later ones ran with accumulate == true by construction).
static int wavefront_trace(lpt_renderer *r, const float view[16], uint32_t n_samples, uint32_t frame_count0, uint32_t seed0, bool acc0,
                           uint32_t slot0, uint32_t piece_slots, Ticket &tk, bool solo) {
    HIP_TRY(hipSetDevice(r->dev->ordinal));
    hipStream_t sm = r->stream;                  // accumulation, filter passes, bookkeeping, reads, the exchange: in call order
    const uint32_t nb = r->max_bounces;          // reference constant 3 (:398-399)

    FrameParams p;
    p.right = mk3(view[0], view[1], view[2]);    // camera.rs:101-108: cols = right, up, direction, origin
    p.up = mk3(view[4], view[5], view[6]);
    p.fwd = mk3(view[8], view[9], view[10]);
    p.origin = mk3(view[12], view[13], view[14]);
    const float th = tanf(0.5f * r->vfov);
    const float aspect = (float)r->w / (float)r->h;
    p.ax = aspect * th; p.ay = th;
    p.width = r->w; p.height = r->h;             // camera.dimensions / global_uniforms.dimensions (:431,436)
    p.user_seed = r->user_seed;
    p.seed_counter = seed0;
    p.rank = r->rank; p.world = r->world; p.tile_w = r->tile_w; p.tile_h = r->tile_h;
    p.map = r->map;
    shard_geometry(r, p.tiles_x, p.n_tiles, p.n_slots);
    p.slot0 = slot0;
    p.block8 = (r->tile_w % 8u == 0u && r->tile_h % 8u == 0u) ? 1u : 0u;
    p.n_slots = piece_slots;                     // the slots of THIS wavefront
    p.frame_count = frame_count0;
    p.max_bounces = nb;
    p.n_samples = n_samples;
    p.fc_inc0 = acc0 ? 1u : 0u;
    const uint32_t n_rays = p.n_slots * n_samples;
    const bool denoise = r->mode != LPT_BLIT_PATHTRACE;
    // the lane (Wavefront) this call's rays live in: consecutive calls take the lanes in turn; the denoising modes
    // carry frame-to-frame state (G-buffer ping-pong, motion) and stay on lane 0
    const bool split = r->n_lanes > 1;           // the wavefront runs on the lane's own stream (a single wavefront on the renderer's own stream instead: 2.89 against 2.90 ms per 1/8-shard frame, round 5)
    const int lane = (denoise || !split) ? 0 : (int)(r->lane_rr++ % (uint32_t)r->n_lanes);
    {
        int st = ensure_lane(r, lane);
        if (st != LPT_OK) return st;
    }
    Wavefront &wf = r->wf[lane];
    hipStream_t s = split ? wf.stream : sm;
    r->last_lane = lane;
    if ((size_t)n_rays > wf.ray_cap && p.n_slots) {
        HIP_TRY(hipStreamSynchronize(sm));       // everything that read this lane's buffers has been enqueued behind `sm`'s waits
        HIP_TRY(hipStreamSynchronize(s));
        int st = alloc_ray_buffers(wf, n_rays);
        if (st != LPT_OK) return st;
        // the next wavefronts of this size take the other lanes: give them their buffers now, not in the middle of a later frame
        // (a first-use hipMalloc of gigabytes takes tens of milliseconds on some hosts)
        for (int l = 0; split && !denoise && l < r->n_lanes; ++l) {
            if (l == lane || r->wf[l].ray_cap >= (size_t)n_rays) continue;
            st = ensure_lane(r, l);
            if (st != LPT_OK) return st;
            HIP_TRY(hipStreamSynchronize(r->wf[l].stream));
            st = alloc_ray_buffers(r->wf[l], n_rays);
            if (st != LPT_OK) return st;
        }
    }
    GBufArgs gb{};
    if (denoise) {
        int st = ensure_denoiser(r);
        if (st != LPT_OK) return st;
    }
    // the accumulation that consumed this lane's previous radiance.  A denoising frame also overwrites what the previous
    // frame's filter passes read (G-buffer ping-pong, motion) — and on a sharded frame those are enqueued by a later call
    // (lpt_renderer_exchange / denoise_filter) — so it waits for everything the renderer's stream holds so far.
    if (split && denoise) { HIP_TRY(hipEventRecord(wf.consumed, sm)); wf.consumed_recorded = true; }
    if (split && wf.consumed_recorded) HIP_TRY(hipStreamWaitEvent(s, wf.consumed, 0));
    if (denoise) {
        r->den_cur = 1 - r->den_cur;         // asvgf.start() (renderer.rs:467)
        if (r->world != 1u) {
            // sharded frame: this rank fills only its tiles; the rest must read as zero for the exchange
            const size_t npx = (size_t)r->w * r->h;
            HIP_TRY(hipMemsetAsync(r->den_gbuf[r->den_cur], 0, sizeof(uint4) * npx, s));
            HIP_TRY(hipMemsetAsync(r->den_motion, 0, sizeof(float2) * npx, s));
            HIP_TRY(hipMemsetAsync(r->den_noisy, 0, sizeof(float4) * npx, sm));
        }
        gb.gbuf = r->den_gbuf[r->den_cur];
        gb.motion = r->den_motion;
        gb.cur.origin = p.origin; gb.cur.right = p.right; gb.cur.up = p.up; gb.cur.fwd = p.fwd; gb.cur.ax = p.ax; gb.cur.ay = p.ay;
        gb.prev = r->prev_cam;
    }

    DProbe probe = r->probe ? r->probe->d : DProbe{(const uint8_t *)r->default_probe, 1u, 1u};
    DNoise nz{(const uint8_t *)r->noise, r->noise_w, r->noise_h, (r->use_noise && r->noise) ? 1u : 0u};
    const DScene &sc = r->sg->d;
    if (r->timings) { r->ring_pos++; harvest_slot(r, cur_slot(r)); }

    if (p.n_slots) {
        HIP_TRY(hipMemsetAsync(wf.ctr, 0, sizeof(FrameCounters), s));
        const uint32_t cus = (uint32_t)r->dev->compute_units;
        const uint32_t stream_blocks = std::min<uint32_t>(div_up(n_rays, kBlock), cus * 8u);
        const uint32_t shade_blocks = std::min<uint32_t>(div_up(n_rays, kBlock), cus * (r->shade_blocks_per_cu ? r->shade_blocks_per_cu : (solo ? 4u : 3u)));
        // the one-round-trip step (kernels.h ray_step_pipe): 72 VGPRs (round 6; 78 before), 6 waves per SIMD used of the 7 that fit, ~3 % more nodes and ~12 % more
        // triangles fetched per ray — and still 1 % less time per frame at 8 M rays, 2 % for a 1 M-ray tile shard
        // (profiles/r03_experiments_ab.txt); LPT_EXP_PIPE_RAYS 0 selects the two-round-trip step
        const bool pipe = n_rays <= r->pipe_rays;
        // persistent waves: about 2.5 primary rays per lane, between 8 waves per CU and all that fit (24 = 6 per SIMD with the one-round-trip step — 28 measured no faster, round 6 —, 32 with the other).  A 1/8
        // tile shard (1 M rays per launch) is best at 24 either way (round 3, span form, two-round-trip step: 8 / 12 / 16 / 24 / 32
        // waves per CU -> 4.41 / 3.89 / 3.62 / 3.51 / 3.54 ms per frame)
        uint32_t waves = r->trace_waves_per_cu ? cus * r->trace_waves_per_cu : std::min(std::max(n_rays / 160u, cus * 8u), cus * (pipe ? 24u : 32u));
        waves = std::max(8u, waves & ~7u);  // whole groups of 8: one chunk head per XCD
        const uint32_t trace_blocks = std::min<uint32_t>(div_up(n_rays, kTraceBlock), waves);
        const size_t lds = stack_bytes(sc);

        // "ray generation" (:444-448)
        stage_begin(r, ST_RAYGEN, s);
        const bool dense = (r->w % r->tile_w == 0u) && (r->h % r->tile_h == 0u);
        if (dense) {
            hipLaunchKernelGGL(k_raygen<true>, dim3(stream_blocks), dim3(kBlock), 0, s, p, nz, wf.q[0], wf.Lsum, wf.ctr);
        } else {
            hipLaunchKernelGGL(k_raygen<false>, dim3(stream_blocks), dim3(kBlock), 0, s, p, nz, wf.q[0], wf.Lsum, wf.ctr);
        }
        stage_end(r, s);

        // Bounce 0 by packet traversal (one tree walk per 8x8-pixel patch) pays while the patch is narrow: at 1920x1080 a packet enters 17.9 nodes
        // for rays that need 14.9 each, and the walk runs at 11.5 Grays/s against 6.2 per ray; at 240x135 the same patch spans eight times the
        // angle, the walk visits several times the nodes, and the packet launch is a third of the frame (0.43 of 1.37 ms; 0.39 ms at 480x270,
        // where the per-ray launch needs 0.13).  LPT_OPT_PACKET_PRIMARY 2 (default): packets up to 1.8 mrad per pixel; 1: always; 0: never.
        const float pixel_rad = 2.0f * th / (float)std::max(r->h, 1u);
        const bool packet = (r->packet_primary == 1u || (r->packet_primary == 2u && pixel_rad <= kPacketMaxPixelRad));
        tk.packet = packet;
        uint32_t seed = seed0;
        // Traversal launches: closest-hit rays of bounce b+1 and shadow rays of bounce b, both produced by shade(b), are traced by
        // ONE persistent launch (k_trace) — nb+1 traversal launches per frame instead of 2*nb.
        // the occluder-cache probe rides with the stats kernels only (kernels.h OccProbe); its table belongs to the renderer
        OccProbe occ{nullptr, 0u, 0.0f};
        if (r->stats && r->occ_table && r->occ_cell > 0.0f) occ = OccProbe{r->occ_table, kOccEntries - 1u, 1.0f / r->occ_cell};
        // the step budget pays where nothing else fills the tail of a launch: a submission that is ONE wavefront (a tile shard, a small frame).  The
        // pieces of a cut batch overlap on the renderer's lanes and hide each other's tails: 1/2 shard as two wavefronts 6.84 ms without, 6.93 with it
        // (not with the stats kernels: a ray dropped at the budget would be missing from the steps-per-ray histogram and its partial node / triangle counts would be
        // counted again by the cooperative kernel's full re-trace — ADVICE r04)
        const bool tail_launch = !r->stats && (solo || r->budget_split) && n_rays <= r->budget_rays;
        // ... the tail finished in place (tail_walk) comes first where it is on: nothing is dropped then, so there is nothing to re-trace
        const uint32_t tail = tail_launch ? std::min(std::min(r->tail_lanes, kTailMax), (uint32_t)std::max(r->refill, 0)) : 0u;
        const uint32_t budget = (tail_launch && !tail && r->step_budget) ? r->step_budget : 0u;
        const size_t tail_lds = tail ? sizeof(uint32_t) * tail_lds_words(r->sg->stats.max_depth) : 0u;
        // a TINY wavefront (fewer rays than the chip has wave slots): the per-bounce launches with EVERY ray traced by a whole wave (k_trace_coop over the queues) — a lane per
        // ray leaves the chip empty and the frame is one chain of dependent steps (64x36, 4 spp: k_path 0.72 ms per frame, the per-lane launches 0.83, this 0.39)
        const bool coop_all = !r->stats && r->coop_rays && n_rays <= r->coop_rays;
        auto trace = [&](int cb, int sb) {
            stage_begin(r, cb >= 0 ? ST_INTERSECT : ST_SHADOW, s);  // :457-464, :493-498
            const Queue qin = wf.q[(uint32_t)(cb < 0 ? 0 : cb) & 1u];
            const int launch_no = cb >= 0 ? cb : (int)nb;   // which strag_count[] this launch fills
            if (coop_all) {
                const size_t clds = sizeof(uint32_t) * coop_stack_entries(kCoopStack, r->sg->stats.max_depth);
                hipLaunchKernelGGL(k_trace_coop<false>, dim3(std::min(2u * n_rays, cus * kCoopWavesPerCu)), dim3(kTraceBlock), clds, s, sc, qin, wf.hits, wf.sq, wf.Lsum, wf.ctr, cb, sb, (const uint32_t *)nullptr, launch_no);
                stage_end(r, s);
                return;
            }
            if (tail) {
                if (pipe) hipLaunchKernelGGL((k_trace<false, true, true>), dim3(trace_blocks), dim3(kTraceBlock), lds + tail_lds, s, TraceKernargs{sc, qin, wf.hits, wf.sq, wf.Lsum, wf.ctr, cb, sb, r->refill, occ, 0u, wf.strag, launch_no, tail});
                else hipLaunchKernelGGL((k_trace<false, false, true>), dim3(trace_blocks), dim3(kTraceBlock), lds + tail_lds, s, TraceKernargs{sc, qin, wf.hits, wf.sq, wf.Lsum, wf.ctr, cb, sb, r->refill, occ, 0u, wf.strag, launch_no, tail});
            } else if (pipe) {
                if (r->stats) hipLaunchKernelGGL((k_trace<true, true>), dim3(trace_blocks), dim3(kTraceBlock), lds + 64u, s, TraceKernargs{sc, qin, wf.hits, wf.sq, wf.Lsum, wf.ctr, cb, sb, r->refill, occ, budget, wf.strag, launch_no, 0u});
                else hipLaunchKernelGGL((k_trace<false, true>), dim3(trace_blocks), dim3(kTraceBlock), lds, s, TraceKernargs{sc, qin, wf.hits, wf.sq, wf.Lsum, wf.ctr, cb, sb, r->refill, occ, budget, wf.strag, launch_no, 0u});
            } else if (r->stats) hipLaunchKernelGGL((k_trace<true, false>), dim3(trace_blocks), dim3(kTraceBlock), lds + 64u, s, TraceKernargs{sc, qin, wf.hits, wf.sq, wf.Lsum, wf.ctr, cb, sb, r->refill, occ, budget, wf.strag, launch_no, 0u});
            else hipLaunchKernelGGL((k_trace<false, false>), dim3(trace_blocks), dim3(kTraceBlock), lds, s, TraceKernargs{sc, qin, wf.hits, wf.sq, wf.Lsum, wf.ctr, cb, sb, r->refill, occ, budget, wf.strag, launch_no, 0u});
            if (budget) {   // the launch's stragglers, a whole wave each (most waves of this grid find none and leave at once)
                const size_t clds = sizeof(uint32_t) * coop_stack_entries(kCoopStack, r->sg->stats.max_depth);
                if (r->stats) hipLaunchKernelGGL(k_trace_coop<true>, dim3(cus * kCoopWavesPerCu), dim3(kTraceBlock), clds, s, sc, qin, wf.hits, wf.sq, wf.Lsum, wf.ctr, cb, sb, wf.strag, launch_no);
                else hipLaunchKernelGGL(k_trace_coop<false>, dim3(cus * kCoopWavesPerCu), dim3(kTraceBlock), clds, s, sc, qin, wf.hits, wf.sq, wf.Lsum, wf.ctr, cb, sb, wf.strag, launch_no);
            }
            stage_end(r, s);
        };
        {
            if (packet) {
                // the primary rays: 64 consecutive queue entries are an 8x8-pixel patch of one sample — packet traversal (k_trace_packet)
                stage_begin(r, ST_PRIMARY, s);
                const uint32_t packets = div_up(n_rays, 64u);
                const size_t plds = (size_t)(48u + 7u * r->sg->stats.max_depth + 8u) * sizeof(uint32_t);   // 48 planes + the stack
                // 4 samples of a 4x4-pixel quarter per packet instead of one sample of an 8x8 patch, where the queue order allows it: a dense frame
                // (queue index = sample * slots + slot), 8x8 pixel blocks inside the tiles, whole blocks, a multiple of four samples
                const uint32_t quad_slots = (r->packet_quads && dense && p.block8 && p.n_slots % 64u == 0u && p.slot0 % 64u == 0u && n_samples % 4u == 0u) ? p.n_slots : 0u;
                if (r->stats) hipLaunchKernelGGL(k_trace_packet<true>, dim3(std::min(packets, cus * kPacketBlocksPerCu)), dim3(kTraceBlock), plds, s, sc, wf.q[0], wf.hits, wf.ctr, 0, quad_slots);
                else hipLaunchKernelGGL(k_trace_packet<false>, dim3(std::min(packets, cus * kPacketBlocksPerCu)), dim3(kTraceBlock), plds, s, sc, wf.q[0], wf.hits, wf.ctr, 0, quad_slots);
                stage_end(r, s);
            } else if (coop_all || !(r->path_rays && n_rays <= r->path_rays)) trace(0, -1);   // a path-kernel wavefront traces its primary rays itself
        }
        // A small wavefront (the tile shard of a multi-GPU frame): every bounce behind the primary hits in ONE persistent launch — the
        // passes of renderer.rs:484-509 without a chip-wide barrier between them (kernels.h k_path); same frame, same counters
        const bool path = r->path_rays && n_rays <= r->path_rays && !coop_all;   // the primary hits are there, whichever kernel found them
        if (path) {
            stage_begin(r, ST_PATH, s);
            const uint32_t pblocks = std::min<uint32_t>(div_up(n_rays, kTraceBlock), std::max(8u, (cus * r->path_waves_per_cu) & ~7u));
            const size_t plds = lds + kPathLdsExtra;   // stacks + sRGB table + per-bounce counters
            if (denoise) {
                if (r->stats) hipLaunchKernelGGL((k_path<true, true>), dim3(pblocks), dim3(kTraceBlock), plds, s, sc, probe, nz, p, wf.q[0], packet ? wf.hits : (const float4 *)nullptr, wf.Lsum, wf.ctr, seed0, gb, r->path_refill);
                else hipLaunchKernelGGL((k_path<true, false>), dim3(pblocks), dim3(kTraceBlock), plds, s, sc, probe, nz, p, wf.q[0], packet ? wf.hits : (const float4 *)nullptr, wf.Lsum, wf.ctr, seed0, gb, r->path_refill);
            } else if (r->stats) hipLaunchKernelGGL((k_path<false, true>), dim3(pblocks), dim3(kTraceBlock), plds, s, sc, probe, nz, p, wf.q[0], packet ? wf.hits : (const float4 *)nullptr, wf.Lsum, wf.ctr, seed0, gb, r->path_refill);
            else hipLaunchKernelGGL((k_path<false, false>), dim3(pblocks), dim3(kTraceBlock), plds, s, sc, probe, nz, p, wf.q[0], packet ? wf.hits : (const float4 *)nullptr, wf.Lsum, wf.ctr, seed0, gb, r->path_refill);
            stage_end(r, s);
        }
        for (uint32_t b = 0; b < nb && !path; ++b) {
            seed += 1u;                          // :453, :487
            const Queue qin = wf.q[b & 1u], qout = wf.q[(b + 1u) & 1u];
            stage_begin(r, ST_SHADE, s);            // :471-480, :502-508
            if (denoise && b == 0u)  // PrimaryRayPass: bounce-0 shading + G-buffer + motion (renderer.rs:466-481)
                hipLaunchKernelGGL(k_shade<true>, dim3(shade_blocks), dim3(kBlock), 0, s, sc, probe, nz, p, qin, wf.hits, qout, wf.sq, wf.Lsum, wf.ctr, (int)b, seed, gb, r->sort_queues);
            else
                hipLaunchKernelGGL(k_shade<false>, dim3(shade_blocks), dim3(kBlock), 0, s, sc, probe, nz, p, qin, wf.hits, qout, wf.sq, wf.Lsum, wf.ctr, (int)b, seed, gb, r->sort_queues);
            stage_end(r, s);
            trace(b + 1u < nb ? (int)(b + 1u) : -1, (int)b);
        }
        if (split) HIP_TRY(hipEventRecord(wf.done, s));
        HIP_TRY(hipGetLastError());
    }
    tk.p = p; tk.lane = lane; tk.split = split; tk.denoise = denoise; tk.cur = gb.cur;
    return LPT_OK;
}

// Second half, on the renderer's stream behind every earlier call's: what reads the wavefront's radiance — accumulation (or the
// denoiser's inputs and filter) and the bookkeeping.  `read` (world == 1, Pathtrace): the pixel rows [row0, row1) this wavefront
// completes are resolved (or tonemapped) and copied to the host right behind its accumulation.
static int wavefront_finish(lpt_renderer *r, const Ticket &tk, const ReadPlan *read, uint32_t row0, uint32_t row1) {
    hipStream_t sm = r->stream;
    const FrameParams &p = tk.p;
    Wavefront &wf = r->wf[tk.lane];
    const bool split = tk.split;
    const uint32_t nb = p.max_bounces;
    if (p.n_slots) {
        const uint32_t stream_blocks = std::min<uint32_t>(div_up(p.n_slots * p.n_samples, kBlock), (uint32_t)r->dev->compute_units * 8u);
        if (split) HIP_TRY(hipStreamWaitEvent(sm, wf.done, 0));
        if (r->mode == LPT_BLIT_PATHTRACE) {
            // AccumulationPass (:523-538)
            stage_begin(r, ST_ACCUM, sm);
            hipLaunchKernelGGL(k_accumulate, dim3(stream_blocks), dim3(kBlock), 0, sm, p, wf.Lsum, r->accum);
            stage_end(r, sm);
            if (read && row1 > row0) {
                // these pixel rows are final: their read-back overlaps the wavefronts that are still tracing the other rows
                const size_t off = (size_t)row0 * r->w, cnt = (size_t)(row1 - row0) * r->w;
                if (read->radiance) {
                    hipLaunchKernelGGL(k_resolve, dim3(div_up((uint32_t)cnt, kBlock)), dim3(kBlock), 0, sm, r->accum + off, r->scratch + off, (uint32_t)cnt);
                    HIP_TRY(hipMemcpyAsync(read->radiance + 4u * off, r->scratch + off, sizeof(float4) * cnt, hipMemcpyDeviceToHost, sm));
                } else {
                    uchar4 *px = reinterpret_cast<uchar4 *>(r->scratch) + off;
                    hipLaunchKernelGGL(k_tonemap, dim3(div_up((uint32_t)cnt, kBlock)), dim3(kBlock), 0, sm, r->accum + off, px, (uint32_t)cnt, r->srgb_thr);
                    HIP_TRY(hipMemcpy2DAsync(read->rgba8 + (size_t)row0 * read->row_bytes, read->row_bytes, px, (size_t)r->w * 4, (size_t)r->w * 4, row1 - row0, hipMemcpyDeviceToHost, sm));
                }
            }
        } else {
            // per-pixel filter inputs; on a sharded frame (world > 1) the caller now exchanges noisy / gbuffer / motion
            // (lpt_renderer_denoiser_inputs) and rank 0 calls lpt_renderer_denoise_filter
            stage_begin(r, ST_ASVGF, sm);
            hipLaunchKernelGGL(k_den_scatter, dim3(stream_blocks), dim3(kBlock), 0, sm, p, wf.Lsum, r->den_noisy);
            r->den_inputs_ready = true;
            if (r->world == 1u) launch_filter(r, sm);
            stage_end(r, sm);
        }  // GBuffer / MotionVector: the primary pass has written the debug targets; nothing else runs (:539)
        hipLaunchKernelGGL(k_finish_frame, dim3(1), dim3(64), 0, sm, wf.ctr, r->totals, nb, tk.packet ? 1u : 0u);
        if (split) {
            HIP_TRY(hipEventRecord(wf.consumed, sm));
            wf.consumed_recorded = true;
        }
        HIP_TRY(hipGetLastError());
    } else if (r->mode != LPT_BLIT_PATHTRACE) {
        r->den_inputs_ready = true;   // a rank without tiles (a pure compositor / filter rank) still takes part in the frame's exchange
    }
    if (tk.denoise) r->prev_cam = tk.cur;        // prev_model_to_screen = P * V^-1 (:542-546)
    return LPT_OK;
}

// Submit what has been recorded (every synchronisation point and every setter that the launches read calls this first).
// With an explicit batch size (lpt_renderer_set_max_fused(n >= 1), lpt_renderer_raytrace_n) the recorded calls are ONE wavefront.
// Otherwise a batch of more than about 4 M rays is cut SPATIALLY — runs of whole tile rows (of whole tiles on a sharded frame),
// every run with all the recorded samples — and the pieces take the renderer's lanes in turn: at 1920x1080 the 4 samples of a
// frame leave as two wavefronts of 4.1 M rays (the upper and the lower half of the image), the shading of one overlaps the
// traversal of the other (measured for two 4 M-ray wavefronts: 13.30 ms per frame against 13.68 for one 8 M-ray wavefront and
// 14.1 for four 2 M-ray ones), and the piece that was launched first is complete about one stage before the last:
// `read` (lpt_renderer_read_radiance, lpt_renderer_blit_rgba8) has every piece's rows copied to the host as soon as they are final.
static int flush_pending(lpt_renderer *r, const ReadPlan *read) {
    if (!r->pend.n) return LPT_OK;
    const lpt_renderer::Pending b = r->pend;
    r->pend.n = 0;
    uint32_t tiles_x, n_tiles, n_slots;
    shard_geometry(r, tiles_x, n_tiles, n_slots);
    const uint32_t area = r->tile_w * r->tile_h;
    const bool whole_rows = r->world == 1u;                       // the rank's slots are the tiles in row-major order
    const uint32_t granule = whole_rows ? tiles_x * area : area;  // slots per tile row / per tile
    const uint32_t granules = granule ? n_slots / granule : 0u;
    uint32_t per_piece = granules;
    // ... and a batch of more than 3 M rays that would still fit one wavefront leaves as TWO (on the renderer's two lanes): a 1/2 shard of the headline frame
    // (4.15 M rays) 7.22 -> 6.81 ms; below that the halves are too small to hide each other's drains (a 1/4 shard, 2 x 1.04 M: 4.62 -> 4.75 ms; profiles/r04_experiments_ab.txt E)
    const uint64_t total = (uint64_t)n_slots * b.n;
    if (!r->max_fused && r->mode == LPT_BLIT_PATHTRACE && granules > 1u && (total > r->wavefront_rays || (r->split_rays && total > r->split_rays && r->n_lanes > 1))) {
        const uint64_t fit = r->wavefront_rays / ((uint64_t)granule * b.n);           // granules of b.n samples in about 4 M rays
        const uint32_t pieces = std::max(2u, div_up(granules, (uint32_t)std::max<uint64_t>(fit, 1u)));
        per_piece = div_up(granules, pieces);                     // the same number of pieces, evened out
    }
    Ticket tk[kMaxLanes];
    if (!granules) {   // nothing owned (a compositor rank): one empty wavefront keeps the bookkeeping of the call
        r->n_wavefronts++;
        const int st = wavefront_trace(r, b.view, b.n, b.frame_count0, b.seed0, b.acc0, 0u, 0u, tk[0], true);
        return st != LPT_OK ? st : wavefront_finish(r, tk[0], nullptr, 0u, 0u);
    }
    // The first halves run ahead of the second halves by the number of lanes: wavefront k's launches are enqueued before the
    // renderer's stream is given the accumulation (and the read-back copy, which blocks the host when the destination is pageable
    // memory) of wavefront k - lanes — the wavefront that used the same lane, whose buffers k overwrites.
    const bool early = read && whole_rows && r->mode == LPT_BLIT_PATHTRACE;
    const uint32_t ahead = (r->mode != LPT_BLIT_PATHTRACE || r->n_lanes < 2) ? 1u : (uint32_t)r->n_lanes;
    const uint32_t pieces = div_up(granules, per_piece);
    auto finish = [&](uint32_t k) {
        const uint32_t g0 = k * per_piece, g1 = std::min(granules, g0 + per_piece);
        return wavefront_finish(r, tk[k % ahead], early ? read : nullptr, std::min(r->h, g0 * r->tile_h), std::min(r->h, g1 * r->tile_h));
    };
    // A failure part way leaves wavefronts on the lanes' streams whose second halves (the waits that put them behind the renderer's
    // stream) were never enqueued: wait for them here, so that nothing is still running on buffers a later call frees or reuses
    auto bail = [&](int st) {
        for (int l = 0; l < kMaxLanes; ++l) {
            if (r->wf[l].stream) hipStreamSynchronize(r->wf[l].stream);
            r->wf[l].consumed_recorded = false;
        }
        hipStreamSynchronize(r->stream);
        return st;
    };
    for (uint32_t k = 0; k < pieces; ++k) {
        if (k >= ahead) { const int st = finish(k - ahead); if (st != LPT_OK) return bail(st); }
        const uint32_t g0 = k * per_piece, g1 = std::min(granules, g0 + per_piece);
        r->n_wavefronts++;
        const int st = wavefront_trace(r, b.view, b.n, b.frame_count0, b.seed0, b.acc0, g0 * granule, (g1 - g0) * granule, tk[k % ahead], pieces == 1u);
        if (st != LPT_OK) return bail(st);
    }
    for (uint32_t k = pieces > ahead ? pieces - ahead : 0u; k < pieces; ++k) { const int st = finish(k); if (st != LPT_OK) return bail(st); }
    return LPT_OK;
}

// recorded calls that may wait for one submission
static uint32_t fuse_cap(const lpt_renderer *r) { return r->max_fused ? r->max_fused : 64u; }

// Records ONE raytrace() call: the host-side protocol of Renderer::raytrace moves now (frame_back :401, seed :453/:487,
// frame_count :535-537), the launches wait for the next submission point.  A call fuses with the recorded ones when it
// continues the same accumulation from the same view — then the n calls are bit for bit the n samples of one wavefront
// (lpt_renderer_raytrace_n's contract; tests/test_gpu_deferred.py).
static int record_call(lpt_renderer *r, const float view[16]) {
    r->frame_back = !r->frame_back;              // renderer.rs:401
    if (!r->resources_set || !r->sg) return LPT_OK;  // :403-407, :419-422
    if (!r->w || !r->h) return LPT_OK;
    r->presented = false;                        // a new sample: the exchanged frame (if any) is stale
    lpt_renderer::Pending &b = r->pend;
    const bool pathtrace = r->mode == LPT_BLIT_PATHTRACE;
    if (b.n) {
        const uint32_t expect_fc = b.frame_count0 + (b.acc0 ? 1u : 0u) + (b.n - 1u);
        const bool fuses = pathtrace && r->accumulate && r->frame_count == expect_fc && b.n < fuse_cap(r) && memcmp(view, b.view, sizeof b.view) == 0;
        if (!fuses) FLUSH_OR_RETURN(r);
    }
    if (!b.n) {
        memcpy(b.view, view, sizeof b.view);
        b.frame_count0 = r->frame_count; b.seed0 = r->seed; b.acc0 = r->accumulate;
    }
    b.n++;
    r->n_recorded++;
    r->seed += r->max_bounces;                   // seed += 1 per intersect stage, never reset
    if (pathtrace && r->accumulate) r->frame_count += 1u;   // frame_count only moves in the Pathtrace arm (:523-538)
    // the denoising modes carry frame-to-frame state (one temporal pass per call): they launch at once; so does a full batch
    if (!pathtrace || b.n >= fuse_cap(r)) return flush_pending(r);
    return LPT_OK;
}

int lpt_renderer_raytrace(lpt_renderer *r, const float view[16]) {
    if (!r || !view) return fail(LPT_ERR_INVALID_ARG, "lpt_renderer_raytrace: null");
    return record_call(r, view);
}

int lpt_renderer_submit(lpt_renderer *r) {
    if (!r) return fail(LPT_ERR_INVALID_ARG, "lpt_renderer_submit: null");
    return flush_pending(r);
}

int lpt_renderer_get_submission_stats(const lpt_renderer *r, uint64_t *recorded_calls, uint64_t *wavefronts, uint32_t *pending_calls) {
    if (!r) return fail(LPT_ERR_INVALID_ARG, "lpt_renderer_get_submission_stats: null");
    if (recorded_calls) *recorded_calls = r->n_recorded;
    if (wavefronts) *wavefronts = r->n_wavefronts;
    if (pending_calls) *pending_calls = r->pend.n;
    return LPT_OK;
}

int lpt_renderer_set_max_fused(lpt_renderer *r, uint32_t n) {
    if (!r || n > 64u) return fail(LPT_ERR_INVALID_ARG, "lpt_renderer_set_max_fused: 0 (auto) or 1..64");
    FLUSH_OR_RETURN(r);
    r->max_fused = n;
    return LPT_OK;
}

int lpt_renderer_raytrace_n(lpt_renderer *r, const float view[16], uint32_t n_samples) {
    if (!r || !view) return fail(LPT_ERR_INVALID_ARG, "lpt_renderer_raytrace: null");
    if (n_samples == 0u || n_samples > 64u) return fail(LPT_ERR_INVALID_ARG, "lpt_renderer_raytrace_n: n must be in [1,64]");
    // n x { raytrace(view); accumulate = true (app.rs:318); }, submitted at once: the explicit batched form
    const uint32_t keep = r->max_fused;
    if (r->mode == LPT_BLIT_PATHTRACE) r->max_fused = std::min(64u, std::max(fuse_cap(r), n_samples + r->pend.n));   // the caller asked for this batch size
    int st = LPT_OK;
    for (uint32_t k = 0; k < n_samples && st == LPT_OK; ++k) {
        st = record_call(r, view);
        if (n_samples > 1u) r->accumulate = true;
    }
    if (st == LPT_OK) st = flush_pending(r);   // still with the explicit batch size: one wavefront, not the automatic spatial cut
    r->max_fused = keep;
    return st;
}

int lpt_renderer_stream(lpt_renderer *r, void **stream) {
    if (!r || !stream) return fail(LPT_ERR_INVALID_ARG, "lpt_renderer_stream: null");
    FLUSH_OR_RETURN(r);
    *stream = (void *)r->stream;
    return LPT_OK;
}

int lpt_renderer_synchronize(lpt_renderer *r) {
    if (!r) return fail(LPT_ERR_INVALID_ARG, "lpt_renderer_synchronize: null");
    FLUSH_OR_RETURN(r);
    HIP_TRY(hipSetDevice(r->dev->ordinal));
    HIP_TRY(hipStreamSynchronize(r->stream));
    return check_device_error(r);
}

int lpt_renderer_radiance_device_ptr(lpt_renderer *r, void **ptr, size_t *bytes) {
    if (!r || !ptr) return fail(LPT_ERR_INVALID_ARG, "lpt_renderer_radiance_device_ptr: null");
    FLUSH_OR_RETURN(r);
    *ptr = r->accum;
    if (bytes) *bytes = sizeof(float4) * (size_t)r->w * r->h;
    return LPT_OK;
}

int lpt_renderer_read_radiance(lpt_renderer *r, float *dst) {
    if (!r || !dst) return fail(LPT_ERR_INVALID_ARG, "lpt_renderer_read_radiance: null");
    if (r->pend.n && r->accum && r->world == 1u && r->mode == LPT_BLIT_PATHTRACE) {
        // the frame is still recorded: submit it with its own read-back — every wavefront's pixel rows travel to the host as
        // soon as they are final, under the wavefronts that are still tracing (the recorded calls clear `presented`, so the
        // local target is what is shown)
        ReadPlan plan;
        plan.radiance = dst;
        const int st = flush_pending(r, &plan);
        if (st != LPT_OK) return st;
        const hipError_t se = hipStreamSynchronize(r->stream);
        if (se != hipSuccess) return fail(LPT_ERR_READBACK, "failed to read pixels from GPU to CPU: %s", hipGetErrorString(se));
        return check_device_error(r);
    }
    FLUSH_OR_RETURN(r);
    if (!r->accum) return fail(LPT_ERR_READBACK, "failed to read pixels from GPU to CPU: no render target");
    hipError_t e = hipSetDevice(r->dev->ordinal);
    const uint32_t n = r->w * r->h;
    if (e == hipSuccess) { hipLaunchKernelGGL(k_resolve, dim3(div_up(n, kBlock)), dim3(kBlock), 0, r->stream, presented_target(r), r->scratch, n); e = hipGetLastError(); }
    if (e == hipSuccess) e = hipMemcpyAsync(dst, r->scratch, sizeof(float4) * (size_t)n, hipMemcpyDeviceToHost, r->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(r->stream);
    if (e != hipSuccess) return fail(LPT_ERR_READBACK, "failed to read pixels from GPU to CPU: %s", hipGetErrorString(e));
    return check_device_error(r);
}

// Host-side gather of a tile-sharded frame: this rank's OWNED pixels (mean radiance) straight into `frame_dst`, a whole-frame destination in page-locked
// host memory (lpt_host_alloc, or any memory passed to lpt_host_register — e.g. a shared-memory segment every rank of the node maps).  Pixels of other ranks
// are not touched.  Blocking.
int lpt_renderer_read_radiance_owned(lpt_renderer *r, float *frame_dst) {
    if (!r || !frame_dst) return fail(LPT_ERR_INVALID_ARG, "lpt_renderer_read_radiance_owned: null");
    FLUSH_OR_RETURN(r);
    if (!r->accum) return fail(LPT_ERR_READBACK, "failed to read pixels from GPU to CPU: no render target");
    if (r->mode != LPT_BLIT_PATHTRACE) return fail(LPT_ERR_INVALID_ARG, "lpt_renderer_read_radiance_owned: BlitMode::Pathtrace only (the denoising modes filter the whole frame on rank 0)");
    HIP_TRY(hipSetDevice(r->dev->ordinal));
    void *mapped = nullptr;
    if (hipHostGetDevicePointer(&mapped, frame_dst, 0) != hipSuccess || !mapped) {
        (void)hipGetLastError();
        return fail(LPT_ERR_INVALID_ARG, "lpt_renderer_read_radiance_owned: the destination must be page-locked host memory (lpt_host_alloc / lpt_host_register)");
    }
    const FrameParams p = shard_params(r);
    if (p.n_slots) hipLaunchKernelGGL(k_resolve_owned, dim3(stream_grid(r, p.n_slots)), dim3(kBlock), 0, r->stream, p, r->accum, reinterpret_cast<float4 *>(mapped));
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(r->stream);
    if (e != hipSuccess) return fail(LPT_ERR_READBACK, "failed to read pixels from GPU to CPU: %s", hipGetErrorString(e));
    return check_device_error(r);
}

int lpt_host_register(void *ptr, size_t bytes) {
    if (!ptr || !bytes) return fail(LPT_ERR_INVALID_ARG, "lpt_host_register: null");
    HIP_TRY(hipHostRegister(ptr, bytes, hipHostRegisterMapped | hipHostRegisterPortable));
    return LPT_OK;
}
int lpt_host_unregister(void *ptr) {
    if (!ptr) return LPT_OK;
    HIP_TRY(hipHostUnregister(ptr));
    return LPT_OK;
}

int lpt_host_alloc(size_t bytes, void **out) {
    if (!out || !bytes) return fail(LPT_ERR_INVALID_ARG, "lpt_host_alloc: null / zero size");
    HIP_TRY(hipHostMalloc(out, bytes, hipHostMallocDefault));
    return LPT_OK;
}
int lpt_host_free(void *ptr) {
    if (ptr) HIP_TRY(hipHostFree(ptr));
    return LPT_OK;
}

int lpt_renderer_blit_rgba8(lpt_renderer *r, uint8_t *dst, size_t row_bytes) {
    if (!r || !dst || row_bytes < (size_t)r->w * 4) return fail(LPT_ERR_INVALID_ARG, "lpt_renderer_blit_rgba8: bad arguments");
    if (r->pend.n && r->accum && r->world == 1u && r->mode == LPT_BLIT_PATHTRACE) {
        // the frame is still recorded: submit it with its own read-back, piece by piece (lpt_renderer_read_radiance does the same)
        ReadPlan plan;
        plan.rgba8 = dst; plan.row_bytes = row_bytes;
        const int st = flush_pending(r, &plan);
        if (st != LPT_OK) return st;
        const hipError_t se = hipStreamSynchronize(r->stream);
        if (se != hipSuccess) return fail(LPT_ERR_READBACK, "failed to read pixels from GPU to CPU: %s", hipGetErrorString(se));
        return check_device_error(r);
    }
    FLUSH_OR_RETURN(r);
    if (!r->accum) return fail(LPT_ERR_READBACK, "failed to read pixels from GPU to CPU: no render target");
    hipError_t e = hipSetDevice(r->dev->ordinal);
    const uint32_t n = r->w * r->h;
    if (e == hipSuccess) {
        if ((r->mode == LPT_BLIT_GBUFFER || r->mode == LPT_BLIT_MOTION) && r->den_temp)  // debug views (renderer.rs:574-586)
            hipLaunchKernelGGL(k_debug_view, dim3(div_up(n, kBlock)), dim3(kBlock), 0, r->stream, r->den_gbuf[r->den_cur], r->den_motion,
                               (uchar4 *)r->scratch, (int)r->w, (int)r->h, r->mode);
        else
            hipLaunchKernelGGL(k_tonemap, dim3(div_up(n, kBlock)), dim3(kBlock), 0, r->stream, presented_target(r), (uchar4 *)r->scratch, n, r->srgb_thr);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpy2DAsync(dst, row_bytes, r->scratch, (size_t)r->w * 4, (size_t)r->w * 4, r->h, hipMemcpyDeviceToHost, r->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(r->stream);
    if (e != hipSuccess) return fail(LPT_ERR_READBACK, "failed to read pixels from GPU to CPU: %s", hipGetErrorString(e));
    return check_device_error(r);
}

int lpt_renderer_read_pixels(lpt_renderer *r, uint8_t *dst) {
    if (!r || !dst) return fail(LPT_ERR_INVALID_ARG, "lpt_renderer_read_pixels: null");
    const int mode = r->mode;  // read_pixels always reads `main` (renderer.rs:769), whatever the blit mode
    if (mode == LPT_BLIT_GBUFFER || mode == LPT_BLIT_MOTION) r->mode = LPT_BLIT_PATHTRACE;
    const int st = lpt_renderer_blit_rgba8(r, dst, (size_t)r->w * 4);
    r->mode = mode;
    return st;
}

int lpt_renderer_read_denoiser(lpt_renderer *r, uint32_t *gbuffer, float *motion, float *radiance, uint32_t *history) {
    if (!r) return fail(LPT_ERR_INVALID_ARG, "lpt_renderer_read_denoiser: null");
    FLUSH_OR_RETURN(r);
    if (!r->den_temp) return fail(LPT_ERR_READBACK, "failed to read pixels from GPU to CPU: no denoising frame has been traced");
    HIP_TRY(hipSetDevice(r->dev->ordinal));
    HIP_TRY(hipStreamSynchronize(r->stream));
    const size_t n = (size_t)r->w * r->h;
    const int c = r->den_cur;
    if (gbuffer) HIP_TRY(hipMemcpy(gbuffer, r->den_gbuf[c], sizeof(uint4) * n, hipMemcpyDeviceToHost));
    if (motion) HIP_TRY(hipMemcpy(motion, r->den_motion, sizeof(float2) * n, hipMemcpyDeviceToHost));
    if (radiance) HIP_TRY(hipMemcpy(radiance, r->den_rad[c], sizeof(float4) * n, hipMemcpyDeviceToHost));
    if (history) HIP_TRY(hipMemcpy(history, r->den_hist[c], sizeof(uint32_t) * n, hipMemcpyDeviceToHost));
    return LPT_OK;
}

// Sharded frames (set_shard, world > 1) in a denoising BlitMode: raytrace() leaves this rank's part of the filter inputs —
// noisy radiance (float4), G-buffer (uint4), motion (float2), zero outside its tiles — in full-frame buffers.  The host
// sums them over the ranks (one reduce each; the tiles are disjoint, so the sum is a gather) into rank 0's buffers and
// calls lpt_renderer_denoise_filter there, which runs the temporal / a-trous / composite passes over the whole frame.
int lpt_renderer_denoiser_inputs(lpt_renderer *r, void **noisy, void **gbuffer, void **motion, size_t *n_pixels) {
    if (!r) return fail(LPT_ERR_INVALID_ARG, "lpt_renderer_denoiser_inputs: null");
    FLUSH_OR_RETURN(r);
    if (!r->den_temp) return fail(LPT_ERR_INVALID_ARG, "lpt_renderer_denoiser_inputs: no denoising frame has been traced");
    if (noisy) *noisy = r->den_noisy;
    if (gbuffer) *gbuffer = r->den_gbuf[r->den_cur];
    if (motion) *motion = r->den_motion;
    if (n_pixels) *n_pixels = (size_t)r->w * r->h;
    return LPT_OK;
}

int lpt_renderer_denoise_filter(lpt_renderer *r) {
    if (!r) return fail(LPT_ERR_INVALID_ARG, "lpt_renderer_denoise_filter: null");
    FLUSH_OR_RETURN(r);
    if (!r->den_temp || !r->den_inputs_ready) return fail(LPT_ERR_INVALID_ARG, "lpt_renderer_denoise_filter: no denoising frame has been traced");
    if (r->world == 1u) return LPT_OK;  // raytrace() has already filtered the frame
    HIP_TRY(hipSetDevice(r->dev->ordinal));
    launch_filter(r, r->stream);
    HIP_TRY(hipGetLastError());
    r->den_inputs_ready = false;
    return LPT_OK;
}

int lpt_renderer_get_ray_counts(lpt_renderer *r, lpt_ray_counts *out) {
    if (!r || !out) return fail(LPT_ERR_INVALID_ARG, "lpt_renderer_get_ray_counts: null");
    FLUSH_OR_RETURN(r);
    HIP_TRY(hipSetDevice(r->dev->ordinal));
    Totals t;
    HIP_TRY(hipMemcpyAsync(&t, r->totals, sizeof t, hipMemcpyDeviceToHost, r->stream));
    HIP_TRY(hipStreamSynchronize(r->stream));
    out->closest = t.closest; out->shadow = t.shadow; out->shaded = t.shaded; out->nodes = t.nodes; out->tris = t.tris;
    out->shadow_nodes = t.shadow_nodes; out->shadow_tris = t.shadow_tris;
    out->wave_steps = t.wave_steps; out->live_lanes = t.live_lanes; out->node_lanes = t.node_lanes; out->tri_lanes = t.tri_lanes;
    out->primary = t.primary; out->packet_nodes = t.packet_nodes; out->packet_tris = t.packet_tris;
    out->shadow_occluded = t.shadow_occluded; out->occluder_cache_found = t.occ_found; out->occluder_cache_hits = t.occ_hits; out->wave_rays = t.wave_rays;
    return check_device_error(r);
}

int lpt_renderer_reset_ray_counts(lpt_renderer *r) {
    if (!r) return fail(LPT_ERR_INVALID_ARG, "lpt_renderer_reset_ray_counts: null");
    FLUSH_OR_RETURN(r);
    HIP_TRY(hipSetDevice(r->dev->ordinal));
    HIP_TRY(hipMemsetAsync(r->totals, 0, sizeof(Totals), r->stream));
    return LPT_OK;
}

int lpt_renderer_get_timings(lpt_renderer *r, lpt_timing *out, int *inout_count) {
    if (!r || !inout_count) return fail(LPT_ERR_INVALID_ARG, "lpt_renderer_get_timings: null");
    FLUSH_OR_RETURN(r);
    HIP_TRY(hipSetDevice(r->dev->ordinal));
    HIP_TRY(hipStreamSynchronize(r->stream));
    lpt_timing acc[ST_COUNT];
    if (r->ev_start) for (int k = 0; k < lpt_renderer::kRing; ++k) harvest_slot(r, k);
    for (int i = 0; i < ST_COUNT; ++i) {
        memset(&acc[i], 0, sizeof acc[i]);
        snprintf(acc[i].label, sizeof acc[i].label, "%s", kStageLabel[i]);
        acc[i].ms = (float)r->stage_ms[i];
        acc[i].launches = r->stage_launches[i];
    }
    const int n = std::min(*inout_count, (int)ST_COUNT);
    if (out) for (int i = 0; i < n; ++i) out[i] = acc[i];
    *inout_count = ST_COUNT;
    return LPT_OK;
}

int lpt_renderer_get_queue_counts(lpt_renderer *r, uint32_t *closest, uint32_t *shadow, uint32_t cap) {
    if (!r) return fail(LPT_ERR_INVALID_ARG, "lpt_renderer_get_queue_counts: null");
    FLUSH_OR_RETURN(r);
    HIP_TRY(hipSetDevice(r->dev->ordinal));
    const uint32_t n = std::min<uint32_t>(cap, (uint32_t)kMaxBounces);
    HIP_TRY(hipStreamSynchronize(r->stream));
    const FrameCounters *ctr = r->wf[r->last_lane].ctr;
    if (!ctr) { if (closest) memset(closest, 0, sizeof(uint32_t) * n); if (shadow) memset(shadow, 0, sizeof(uint32_t) * n); return LPT_OK; }
    // FrameCounters keeps (shadow count of bounce b, closest-hit count of bounce b + 1) side by side: unpack
    uint32_t host[2 + 2 * kMaxBounces];
    HIP_TRY(hipMemcpy(host, &ctr->q0, sizeof host, hipMemcpyDeviceToHost));
    for (uint32_t b = 0; b < n; ++b) {
        if (closest) closest[b] = b == 0u ? host[0] : host[2u + 2u * (b - 1u) + 1u];
        if (shadow) shadow[b] = host[2u + 2u * b];
    }
    return LPT_OK;
}

// stats kernels only: traversal steps per ray over the per-bounce traversal launches of the LAST wavefront (its longest ray sets a launch's duration)
int lpt_renderer_get_step_histogram(lpt_renderer *r, uint32_t *max_steps, uint32_t *hist12) {
    if (!r) return fail(LPT_ERR_INVALID_ARG, "lpt_renderer_get_step_histogram: null");
    FLUSH_OR_RETURN(r);
    HIP_TRY(hipSetDevice(r->dev->ordinal));
    HIP_TRY(hipStreamSynchronize(r->stream));
    const FrameCounters *ctr = r->wf[r->last_lane].ctr;
    if (max_steps) *max_steps = 0;
    if (hist12) memset(hist12, 0, sizeof(uint32_t) * 12);
    if (!ctr) return LPT_OK;
    if (max_steps) HIP_TRY(hipMemcpy(max_steps, &ctr->max_steps, sizeof(uint32_t), hipMemcpyDeviceToHost));
    if (hist12) HIP_TRY(hipMemcpy(hist12, ctr->step_hist, sizeof(uint32_t) * 12, hipMemcpyDeviceToHost));
    return LPT_OK;
}

// ============================================================================ multi-GPU frame exchange
int lpt_comm_unique_id(void *out_id) {
    if (!out_id) return fail(LPT_ERR_INVALID_ARG, "lpt_comm_unique_id: null");
    static_assert(sizeof(ncclUniqueId) == LPT_COMM_ID_BYTES, "LPT_COMM_ID_BYTES must match ncclUniqueId");
    const Rccl *nc = rccl();
    if (!nc) return LPT_ERR_RCCL;
    ncclUniqueId id;
    RCCL_TRY(nc->GetUniqueId(&id));
    memcpy(out_id, &id, sizeof id);
    return LPT_OK;
}

int lpt_comm_create(lpt_device *dev, const void *id_bytes, int rank, int world, lpt_comm **out) {
    if (!dev || !id_bytes || !out) return fail(LPT_ERR_INVALID_ARG, "lpt_comm_create: null");
    if (world < 1 || rank < 0 || rank >= world) return fail(LPT_ERR_INVALID_ARG, "lpt_comm_create: rank %d of %d", rank, world);
    const Rccl *nc = rccl();
    if (!nc) return LPT_ERR_RCCL;
    HIP_TRY(hipSetDevice(dev->ordinal));
    ncclUniqueId id;
    memcpy(&id, id_bytes, sizeof id);
    lpt_comm *c = new (std::nothrow) lpt_comm();
    if (!c) return fail(LPT_ERR_INVALID_ARG, "out of host memory");
    c->dev = dev; c->rank = rank; c->world = world;
    ncclResult_t st = nc->CommInitRank(&c->comm, world, id, rank);
    if (st != ncclSuccess) { delete c; return fail(LPT_ERR_RCCL, "ncclCommInitRank(rank %d of %d) failed: %s", rank, world, nc->GetErrorString(st)); }
    *out = c;
    return LPT_OK;
}

int lpt_comm_destroy(lpt_comm *c) {
    if (!c) return LPT_OK;
    hipSetDevice(c->dev->ordinal);
    // renderers still bound to the communicator: their recorded frames leave first, then they are unbound (they keep their shard of the frame; an
    // exchange on them is then the single-GPU no-op instead of a call into a destroyed communicator)
    flush_device(c->dev);
    for (lpt_renderer *r : c->dev->renderers)
        if (r->comm == c) { forget_deferred_exchange(r); hipStreamSynchronize(r->stream); r->comm = nullptr; }
    if (c->comm && rccl()) rccl()->CommDestroy(c->comm);
    delete c;
    return LPT_OK;
}

// rank and size as RCCL itself reports them for the communicator (ncclCommUserRank / ncclCommCount), not the values
// lpt_comm_create was called with: a host (bench.py) uses this to check that the N ranks really joined one communicator
int lpt_comm_info(const lpt_comm *c, int *rank, int *world) {
    if (!c) return fail(LPT_ERR_INVALID_ARG, "lpt_comm_info: null");
    const Rccl *nc = rccl();
    if (!nc) return LPT_ERR_RCCL;
    int rk = -1, n = -1;
    if (c->comm) {
        RCCL_TRY(nc->CommUserRank(c->comm, &rk));
        RCCL_TRY(nc->CommCount(c->comm, &n));
    }
    if (rank) *rank = rk;
    if (world) *world = n;
    return LPT_OK;
}

int lpt_shard_layout_weighted(uint32_t width, uint32_t height, uint32_t tile_w, uint32_t tile_h, uint32_t world, uint32_t rank, const uint32_t *weights,
                              uint32_t *out_slots, uint32_t *out_offset) {
    if (!world || rank >= world || !tile_w || !tile_h) return fail(LPT_ERR_INVALID_ARG, "lpt_shard_layout: bad shard (rank %u of %u, tile %ux%u)", rank, world, tile_w, tile_h);
    const uint32_t n_tiles = div_up(width, tile_w) * div_up(height, tile_h), area = tile_w * tile_h;
    if (!weights) {   // unit weights: the closed form (any world size)
        const uint32_t a = shard_slot_offset(n_tiles, world, area, rank), b = shard_slot_offset(n_tiles, world, area, rank + 1u);
        if (out_slots) *out_slots = b - a;
        if (out_offset) *out_offset = a;
        return LPT_OK;
    }
    if (world > kMaxWorld) return fail(LPT_ERR_INVALID_ARG, "weighted shards: at most %u ranks", kMaxWorld);
    uint32_t sum = 0;
    for (uint32_t q = 0; q < world; ++q) { if (weights[q] > kMaxWeight) return fail(LPT_ERR_INVALID_ARG, "weighted shards: weight %u exceeds %u", weights[q], kMaxWeight); sum += weights[q]; }
    if (sum == 0u || sum > kMaxVirtual) return fail(LPT_ERR_INVALID_ARG, "weighted shards: the weights must sum to 1..%u (got %u)", kMaxVirtual, sum);
    std::vector<uint32_t> off;
    make_shard_table(std::vector<uint32_t>(weights, weights + world), world, n_tiles, area, off);
    if (out_slots) *out_slots = off[rank + 1] - off[rank];
    if (out_offset) *out_offset = off[rank];
    return LPT_OK;
}
int lpt_shard_layout(uint32_t width, uint32_t height, uint32_t tile_w, uint32_t tile_h, uint32_t world, uint32_t rank, uint32_t *out_slots, uint32_t *out_offset) {
    return lpt_shard_layout_weighted(width, height, tile_w, tile_h, world, rank, nullptr, out_slots, out_offset);
}
// which rank owns tile `tile` (row-major over the tile grid) under the rule above: for hosts and tests that check the layout
int lpt_shard_owner(uint32_t world, const uint32_t *weights, uint32_t tile, uint32_t *out_rank) {
    if (!world || !out_rank || (weights && world > kMaxWorld)) return fail(LPT_ERR_INVALID_ARG, "lpt_shard_owner: bad arguments");
    std::vector<uint32_t> wv;
    if (weights) {
        uint32_t sum = 0;
        for (uint32_t q = 0; q < world; ++q) { if (weights[q] > kMaxWeight) return fail(LPT_ERR_INVALID_ARG, "weighted shards: weight %u exceeds %u", weights[q], kMaxWeight); sum += weights[q]; }
        if (sum == 0u || sum > kMaxVirtual) return fail(LPT_ERR_INVALID_ARG, "weighted shards: the weights must sum to 1..%u (got %u)", kMaxVirtual, sum);
        wv.assign(weights, weights + world);
    }
    const std::vector<uint32_t> owner = deal_virtual_ranks(wv, world);
    *out_rank = owner[tile % owner.size()];
    return LPT_OK;
}

int lpt_renderer_set_comm_weighted(lpt_renderer *r, lpt_comm *comm, const uint32_t *weights) {
    if (!r) return fail(LPT_ERR_INVALID_ARG, "lpt_renderer_set_comm: null");
    if (comm && comm->dev != r->dev) return fail(LPT_ERR_INVALID_ARG, "lpt_renderer_set_comm: the communicator belongs to another device");
    FLUSH_OR_RETURN(r);
    // bind only once the shard is in place: a failed set_shard must not leave a half-bound renderer
    const int st = comm ? lpt_renderer_set_shard_weighted(r, (uint32_t)comm->rank, (uint32_t)comm->world, 32u, 8u, weights) : lpt_renderer_set_shard(r, 0u, 1u, 32u, 8u);
    if (st == LPT_OK) r->comm = comm;
    return st;
}
int lpt_renderer_set_comm(lpt_renderer *r, lpt_comm *comm) { return lpt_renderer_set_comm_weighted(r, comm, nullptr); }

}  // extern "C"

// the sharding half of FrameParams (what k_pack_owned / k_unpack_frame read)
static FrameParams shard_params(const lpt_renderer *r) {
    FrameParams p{};
    p.width = r->w; p.height = r->h;
    p.rank = r->rank; p.world = r->world; p.tile_w = r->tile_w; p.tile_h = r->tile_h;
    p.map = r->map;
    p.block8 = (r->tile_w % 8u == 0u && r->tile_h % 8u == 0u) ? 1u : 0u;
    shard_geometry(r, p.tiles_x, p.n_tiles, p.n_slots);
    return p;
}

// `frame` (the presented whole frame: only where it is written — the root, or every rank of a reduce, which needs a valid
// receive buffer) and the staging area for packed tiles: this rank's slots, or every rank's on the root
static int ensure_exchange_buffers(lpt_renderer *r, bool root, bool want_frame, bool want_stage, size_t bytes_per_slot = 16) {
    const size_t npx = (size_t)r->w * r->h;
    if ((root || want_frame) && !r->frame) HIP_TRY(hipMalloc(&r->frame, sizeof(float4) * std::max<size_t>(npx, 1)));
    if (!r->xevent) HIP_TRY(hipEventCreateWithFlags(&r->xevent, hipEventDisableTiming));
    if (!want_stage) return LPT_OK;
    const FrameParams p = shard_params(r);
    const size_t slots = root ? (size_t)p.n_tiles * p.tile_w * p.tile_h : (size_t)p.n_slots;
    const size_t need = std::max<size_t>((slots * bytes_per_slot + 15) / 16, 1);   // in float4 units
    if (r->xstage_elems < need) {
        HIP_TRY(hipStreamSynchronize(r->stream));
        if (r->xstage) hipFree(r->xstage);
        r->xstage = nullptr; r->xstage_elems = 0;
        HIP_TRY(hipMalloc(&r->xstage, sizeof(float4) * need));
        r->xstage_elems = need;
    }
    return LPT_OK;
}
static inline uint32_t stream_grid(const lpt_renderer *r, size_t n) {
    return (uint32_t)std::max<size_t>(1, std::min<size_t>((n + kBlock - 1) / kBlock, (size_t)r->dev->compute_units * 8u));
}

// Phase 1 of an exchange: pack + the RCCL operations, enqueued on the renderer's stream.  Inside an open
// lpt_comm_group_begin / _end bracket RCCL only issues them at the outermost ncclGroupEnd, so whatever consumes the received
// data (phase 2: unpack, filter) must be enqueued after that — lpt_comm_group_end runs the deferred phase 2 of every renderer.
static int exchange_enqueue(lpt_renderer *r, int mode) {
    lpt_comm *c = r->comm;
    hipStream_t s = r->stream;
    const bool root = c->rank == 0;
    const size_t npx = (size_t)r->w * r->h;
    const Rccl &nc = *rccl();
    stage_begin(r, ST_EXCHANGE, s);
    if (r->mode != LPT_BLIT_PATHTRACE) {
        // denoising BlitModes: the per-pixel filter inputs are what is exchanged (zero outside a rank's tiles, so the sums
        // are gathers); rank 0 then filters the whole frame (SPEC §15.5)
        if (mode == LPT_EXCHANGE_REDUCE) {
            RCCL_TRY(nc.Reduce(r->den_noisy, r->den_noisy, 4 * npx, ncclFloat32, ncclSum, 0, c->comm, s));
            RCCL_TRY(nc.Reduce(r->den_gbuf[r->den_cur], r->den_gbuf[r->den_cur], 4 * npx, ncclInt32, ncclSum, 0, c->comm, s));
            RCCL_TRY(nc.Reduce(r->den_motion, r->den_motion, 2 * npx, ncclFloat32, ncclSum, 0, c->comm, s));
            return LPT_OK;
        }
        // owned tiles only: 40 B per owned pixel (41 MB per rank for a 3840x2160 frame on 8 GPUs, against 330 MB of reduces)
        int st = ensure_exchange_buffers(r, root, false, true, 40);
        if (st != LPT_OK) return st;
        const FrameParams p = shard_params(r);
        unsigned char *stage = reinterpret_cast<unsigned char *>(r->xstage);
        if (p.n_slots) hipLaunchKernelGGL(k_pack_den, dim3(stream_grid(r, p.n_slots)), dim3(kBlock), 0, s, p, r->den_noisy, r->den_gbuf[r->den_cur], r->den_motion, stage);
        HIP_TRY(hipGetLastError());
        if (root) {
            RCCL_TRY(nc.GroupStart());
            for (uint32_t q = 1; q < p.world; ++q) {
                const uint32_t nq = r->h_offset[q + 1] - r->h_offset[q];
                if (!nq) continue;
                ncclResult_t e = nc.Recv(stage + 40u * (size_t)r->h_offset[q], 40u * (size_t)nq, ncclUint8, (int)q, c->comm, s);
                if (e != ncclSuccess) { nc.GroupEnd(); return fail(LPT_ERR_RCCL, "ncclRecv from rank %u failed: %s", q, nc.GetErrorString(e)); }
            }
            RCCL_TRY(nc.GroupEnd());
        } else if (p.n_slots) {
            RCCL_TRY(nc.Send(stage, 40u * (size_t)p.n_slots, ncclUint8, 0, c->comm, s));
        }
        return LPT_OK;
    }
    int st = ensure_exchange_buffers(r, root, mode == LPT_EXCHANGE_REDUCE, mode == LPT_EXCHANGE_GATHER_TILES);
    if (st != LPT_OK) return st;
    const FrameParams p = shard_params(r);
    if (mode == LPT_EXCHANGE_REDUCE) {
        // every rank passes a valid receive buffer (only the root's is written)
        RCCL_TRY(nc.Reduce(r->accum, r->frame, 4 * npx, ncclFloat32, ncclSum, 0, c->comm, s));
        return LPT_OK;
    }
    if (p.n_slots) hipLaunchKernelGGL(k_pack_owned, dim3(stream_grid(r, p.n_slots)), dim3(kBlock), 0, s, p, r->accum, r->xstage);   // rank 0's offset is 0
    HIP_TRY(hipGetLastError());
    if (root) {
        RCCL_TRY(nc.GroupStart());
        for (uint32_t q = 1; q < p.world; ++q) {
            const uint32_t nq = r->h_offset[q + 1] - r->h_offset[q];
            if (!nq) continue;
            ncclResult_t e = nc.Recv(r->xstage + r->h_offset[q], 4 * (size_t)nq, ncclFloat32, (int)q, c->comm, s);
            if (e != ncclSuccess) { nc.GroupEnd(); return fail(LPT_ERR_RCCL, "ncclRecv from rank %u failed: %s", q, nc.GetErrorString(e)); }
        }
        RCCL_TRY(nc.GroupEnd());
    } else if (p.n_slots) {
        RCCL_TRY(nc.Send(r->xstage, 4 * (size_t)p.n_slots, ncclFloat32, 0, c->comm, s));
    }
    return LPT_OK;
}

// Phase 2: what reads the received data, on the same stream, behind the RCCL operations
static int exchange_finish(lpt_renderer *r, int mode) {
    hipStream_t s = r->stream;
    const bool root = r->comm->rank == 0;
    const size_t npx = (size_t)r->w * r->h;
    if (r->mode != LPT_BLIT_PATHTRACE) {
        if (root && mode == LPT_EXCHANGE_GATHER_TILES) {
            const FrameParams p = shard_params(r);
            hipLaunchKernelGGL(k_unpack_den, dim3(stream_grid(r, npx)), dim3(kBlock), 0, s, p, r->d_table, reinterpret_cast<unsigned char *>(r->xstage), r->den_noisy, r->den_gbuf[r->den_cur], r->den_motion);
        }
        if (root && r->world != 1u) launch_filter(r, s);   // world == 1: raytrace() has filtered already
        stage_end(r, s);
        HIP_TRY(hipGetLastError());
        r->den_inputs_ready = false;
        return LPT_OK;   // the composite has written the local target on rank 0
    }
    if (root && mode == LPT_EXCHANGE_GATHER_TILES) {
        const FrameParams p = shard_params(r);
        hipLaunchKernelGGL(k_unpack_frame, dim3(stream_grid(r, npx)), dim3(kBlock), 0, s, p, r->d_table, r->xstage, r->frame);
    }
    stage_end(r, s);
    HIP_TRY(hipGetLastError());
    r->presented = root;
    return LPT_OK;
}

// one thread drives several communicators inside lpt_comm_group_begin / _end: the second phases wait for the outermost end
struct DeferredFinish { lpt_renderer *r; int mode; };
static thread_local int t_group_depth = 0;
static thread_local std::vector<DeferredFinish> t_deferred;
static void forget_deferred_exchange(lpt_renderer *r) {
    t_deferred.erase(std::remove_if(t_deferred.begin(), t_deferred.end(), [r](const DeferredFinish &d) { return d.r == r; }), t_deferred.end());
}

extern "C" {

int lpt_comm_group_begin(void) {
    const Rccl *nc = rccl();
    if (!nc) return LPT_ERR_RCCL;
    RCCL_TRY(nc->GroupStart());
    ++t_group_depth;
    return LPT_OK;
}
int lpt_comm_group_end(void) {
    const Rccl *nc = rccl();
    if (!nc) return LPT_ERR_RCCL;
    if (t_group_depth <= 0) return fail(LPT_ERR_INVALID_ARG, "lpt_comm_group_end without lpt_comm_group_begin");
    const ncclResult_t e = nc->GroupEnd();
    if (--t_group_depth > 0) { if (e != ncclSuccess) return fail(LPT_ERR_RCCL, "ncclGroupEnd failed: %s", nc->GetErrorString(e)); return LPT_OK; }
    std::vector<DeferredFinish> todo;
    todo.swap(t_deferred);
    if (e != ncclSuccess) return fail(LPT_ERR_RCCL, "ncclGroupEnd failed: %s", nc->GetErrorString(e));
    int st = LPT_OK;
    for (const DeferredFinish &d : todo) {   // the operations are on the streams now: enqueue their consumers
        hipSetDevice(d.r->dev->ordinal);
        const int f = exchange_finish(d.r, d.mode);
        if (st == LPT_OK) st = f;
    }
    return st;
}

int lpt_renderer_exchange(lpt_renderer *r, int mode) {
    if (!r) return fail(LPT_ERR_INVALID_ARG, "lpt_renderer_exchange: null");
    if (mode != LPT_EXCHANGE_GATHER_TILES && mode != LPT_EXCHANGE_REDUCE) return fail(LPT_ERR_INVALID_ARG, "lpt_renderer_exchange: unknown mode %d", mode);
    FLUSH_OR_RETURN(r);
    if (!r->comm) return LPT_OK;   // single GPU: the local target is the frame
    if (!r->w || !r->h || !r->accum) return LPT_OK;
    lpt_comm *c = r->comm;
    if ((uint32_t)c->rank != r->rank || (uint32_t)c->world != r->world)
        return fail(LPT_ERR_INVALID_ARG, "lpt_renderer_exchange: the renderer's shard (%u of %u) is not the communicator's (%d of %d); call lpt_renderer_set_comm again",
                    r->rank, r->world, c->rank, c->world);
    if (r->mode != LPT_BLIT_PATHTRACE && (!r->den_temp || !r->den_inputs_ready)) return fail(LPT_ERR_INVALID_ARG, "lpt_renderer_exchange: no denoising frame has been traced");
    if (!rccl()) return LPT_ERR_RCCL;
    HIP_TRY(hipSetDevice(r->dev->ordinal));
    if (r->timings) { r->ring_pos++; harvest_slot(r, cur_slot(r)); }
    int st = exchange_enqueue(r, mode);
    if (st != LPT_OK) { stage_end(r, r->stream); return st; }   // a failed exchange still closes its timing stage (get_timings after a failure)
    if (t_group_depth > 0) {   // the RCCL operations are only issued by the outermost lpt_comm_group_end: finish there
        t_deferred.push_back(DeferredFinish{r, mode});
        return LPT_OK;
    }
    return exchange_finish(r, mode);
}

int lpt_renderer_exchange_local(lpt_renderer *root, lpt_renderer *const *peers, int n_peers) {
    if (!root || n_peers < 0 || (n_peers && !peers)) return fail(LPT_ERR_INVALID_ARG, "lpt_renderer_exchange_local: null");
    if (root->rank != 0u || root->world != (uint32_t)n_peers + 1u)
        return fail(LPT_ERR_INVALID_ARG, "lpt_renderer_exchange_local: root must be rank 0 of %d (is %u of %u)", n_peers + 1, root->rank, root->world);
    const bool den = root->mode != LPT_BLIT_PATHTRACE;   // denoising BlitModes: the filter inputs travel, rank 0 filters
    FLUSH_OR_RETURN(root);
    for (int i = 0; i < n_peers; ++i) if (peers[i]) FLUSH_OR_RETURN(peers[i]);
    if (!root->w || !root->h || !root->accum) return LPT_OK;
    std::vector<char> seen(root->world, 0);
    seen[0] = 1;
    for (int i = 0; i < n_peers; ++i) {
        const lpt_renderer *q = peers[i];
        if (!q || q->w != root->w || q->h != root->h || q->world != root->world || q->tile_w != root->tile_w || q->tile_h != root->tile_h || q->rank >= root->world || seen[q->rank] ||
            q->weights != root->weights || (q->mode != LPT_BLIT_PATHTRACE) != den)
            return fail(LPT_ERR_INVALID_ARG, "lpt_renderer_exchange_local: peer %d does not complete the shard set of the root", i);
        if (den && (!q->den_temp || !q->den_inputs_ready)) return fail(LPT_ERR_INVALID_ARG, "lpt_renderer_exchange_local: peer %d has not traced a denoising frame", i);
        seen[q->rank] = 1;
    }
    if (den && (!root->den_temp || !root->den_inputs_ready)) return fail(LPT_ERR_INVALID_ARG, "lpt_renderer_exchange_local: no denoising frame has been traced");
    const size_t bps = den ? 40 : 16;
    HIP_TRY(hipSetDevice(root->dev->ordinal));
    int st = ensure_exchange_buffers(root, true, true, true, bps);
    if (st != LPT_OK) return st;
    const FrameParams p0 = shard_params(root);
    unsigned char *stage0 = reinterpret_cast<unsigned char *>(root->xstage);
    if (p0.n_slots) {
        if (den) hipLaunchKernelGGL(k_pack_den, dim3(stream_grid(root, p0.n_slots)), dim3(kBlock), 0, root->stream, p0, root->den_noisy, root->den_gbuf[root->den_cur], root->den_motion, stage0);
        else hipLaunchKernelGGL(k_pack_owned, dim3(stream_grid(root, p0.n_slots)), dim3(kBlock), 0, root->stream, p0, root->accum, root->xstage);
    }
    for (int i = 0; i < n_peers; ++i) {
        lpt_renderer *q = peers[i];
        HIP_TRY(hipSetDevice(q->dev->ordinal));
        st = ensure_exchange_buffers(q, false, false, true, bps);
        if (st != LPT_OK) return st;
        const FrameParams pq = shard_params(q);
        // the root's staging area may still be read by the unpack of its previous exchange
        if (root->xevent_recorded) HIP_TRY(hipStreamWaitEvent(q->stream, root->xevent, 0));
        if (pq.n_slots) {
            if (den) hipLaunchKernelGGL(k_pack_den, dim3(stream_grid(q, pq.n_slots)), dim3(kBlock), 0, q->stream, pq, q->den_noisy, q->den_gbuf[q->den_cur], q->den_motion, reinterpret_cast<unsigned char *>(q->xstage));
            else hipLaunchKernelGGL(k_pack_owned, dim3(stream_grid(q, pq.n_slots)), dim3(kBlock), 0, q->stream, pq, q->accum, q->xstage);
            // the stand-in of ncclSend / ncclRecv inside one process: a (peer) copy into the root's staging area
            HIP_TRY(hipMemcpyPeerAsync(stage0 + bps * (size_t)root->h_offset[pq.rank], root->dev->ordinal, q->xstage, q->dev->ordinal,
                                       bps * (size_t)pq.n_slots, q->stream));
        }
        if (den) q->den_inputs_ready = false;
        HIP_TRY(hipEventRecord(q->xevent, q->stream));
        HIP_TRY(hipSetDevice(root->dev->ordinal));
        HIP_TRY(hipStreamWaitEvent(root->stream, q->xevent, 0));
    }
    if (den) {
        hipLaunchKernelGGL(k_unpack_den, dim3(stream_grid(root, (size_t)root->w * root->h)), dim3(kBlock), 0, root->stream, p0, root->d_table, stage0, root->den_noisy, root->den_gbuf[root->den_cur], root->den_motion);
        launch_filter(root, root->stream);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipEventRecord(root->xevent, root->stream));
        root->xevent_recorded = true;
        root->den_inputs_ready = false;
        return LPT_OK;   // the composite has written the local target
    }
    hipLaunchKernelGGL(k_unpack_frame, dim3(stream_grid(root, (size_t)root->w * root->h)), dim3(kBlock), 0, root->stream, p0, root->d_table, root->xstage, root->frame);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(root->xevent, root->stream));
    root->xevent_recorded = true;
    root->presented = true;
    return LPT_OK;
}

}  // extern "C"
