#!/usr/bin/env python3
"""bench.py — Mrays/s of the path-tracing hot path on N MI355X (one process per GPU).

Workload (BASELINE.json metric: "Mrays/s (+ ms/frame) at 1920x1080, 4 spp, Sponza"):
  synthetic_atrium(seed=2) — the Sponza STAND-IN (the real asset is absent) at the sizes §8d gives it
  (262,144 triangles, 20 x 1024^2 textures = 80 MB of texels, 25 materials) — 1920x1080, 4 spp, path depth 8, camera = the
  reference's start pose.  Inputs (scene, BVH, probe, textures) are resident in HBM before the timed region.

`value` is measured over the span SURVEY §8d defines and crates/standalone issues (app.rs:297-337), on ONE renderer per GPU:
    per FRAME:  reset_accumulation(); accumulate = true; 4 x Renderer::raytrace(view); [N>1: lpt_renderer_exchange];
                read_radiance()  (rank 0; blocking, into page-locked host memory; the other ranks synchronise)
  raytrace() RECORDS (record-then-submit, include/lpt.h): the four calls of a frame are launched by the next submission
  point — read_radiance — as wavefronts of 4 samples per pixel (at 1920x1080: two, the upper and the lower half of the image,
  each read back as soon as it is complete), bit-identical to four separate launches.
  A STEP = FRAMES_PER_STEP (10) such frames, so that the driver's `--steps 20` times about 2.5 s instead of 0.25 s.
  value = (closest-hit + shadow rays traced by all ranks in the K timed steps) / wall time, in Mrays/s, with barrier +
  torch.cuda.synchronize() on both sides and the MAX over ranks.
  N>1: frames shard by interleaved 32x8 pixel tiles (tile id mod N): per-GPU work shrinks as N grows ("strong" scaling of one
  frame).  The exchange is native RCCL inside the library (owned-tile gather to rank 0, or --exchange reduce);
  torch.distributed (gloo) is the control plane only — rendezvous of the 128-byte RCCL ids, barriers, the max over ranks.

Launch: `python bench.py --gpus N ...` starts its own N ranks (child processes with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*)
BEFORE anything touches a GPU, relays rank 0's JSON line and exits non-zero with the failing rank's stderr if a rank dies;
under `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N` (WORLD_SIZE set) it is one of the ranks.

Besides `value` the line carries
  throughput  --pipeline P renderers in flight (own HIP streams and, for N>1, own RCCL communicators), the 4 spp of a frame as
              raytrace_n(view, 4), no read-back: what the hardware sustains when frames overlap (round 2's `value`);
  latency_ms  one frame alone without the read-back;
  roofline    k_trace (the per-ray traversal of bounces 1.. and of every shadow ray): algorithmic bytes per launch / the launch
              time (HIP events on the stream the kernel runs on; one frame at a time, nothing co-running) / 8 TB/s, with the
              PMC-derived limits of profiles/limits.json; roofline.packet: the same for k_trace_packet, which traces bounce 0
              as packets of 64 coherent rays (one tree walk per packet);
  rccl        (N>1) what RCCL itself reports for the communicator, the exchange time per frame (HIP events around
              lpt_renderer_exchange on rank 0), per-rank ray counts, and whether rank 0's presented frame was complete;
  cpu_baseline  the oracle ("port") on the host cores, bounded sample.
"""
import argparse
import json
import os
import shutil
import socket
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")  # read by the HIP runtime at initialisation: one hardware queue per stream
os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")  # single-node control plane; the box's hostname may not resolve
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: what RCCL needs on this pool

WIDTH, HEIGHT, SPP, DEPTH = 1920, 1080, 4, 8
FRAMES_PER_STEP = 10
HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the throughput / latency measurements (experiments)")
    ap.add_argument("--throughput", action="store_true", help="N>1: also run the frames-in-flight leg (several renderers, one RCCL communicator each); on one GPU it always runs")
    ap.add_argument("--width", type=int, default=WIDTH, help="experiments only; the reported config is the default")
    ap.add_argument("--height", type=int, default=HEIGHT)
    ap.add_argument("--spp", type=int, default=SPP)
    ap.add_argument("--frames-per-step", type=int, default=FRAMES_PER_STEP)
    ap.add_argument("--texture-size", type=int, default=1024, help="experiments only (SURVEY §8d: 1024)")
    ap.add_argument("--emulate-shard", type=int, default=0, help="experiments: render only rank 0's tiles of an N-way shard on one GPU")
    ap.add_argument("--force-dist", action="store_true", help="take the N>1 code path (process group, communicator, exchange) even with one rank")
    ap.add_argument("--exchange", choices=["auto", "gather", "reduce", "host"], default="auto",
                    help="frame exchange for N>1.  auto (default): RCCL is brought up under a 60 s watchdog (on a timeout or an error the run continues RCCL-free and says so in "
                         "`rccl.error`), every form that came up gets 5 calibration frames, the timed region uses the fastest (`config.exchange`, `exchange_auto`) — the "
                         "headline cannot be lost to RCCL code that has never run with N > 1.  gather: owned tiles only (W*H/N*16 B per rank, grouped send/recv), reduce: the dense ncclReduce of the accumulation buffer, or "
                         "`host`: no exchange on the GPUs — every rank writes its owned pixels straight into ONE shared-memory frame (lpt_renderer_read_radiance_owned; "
                         "each GPU's 1/N over its own PCIe link), completed by a host-side barrier")
    ap.add_argument("--rccl-timeout", type=float, default=60.0, help="--exchange auto: seconds the RCCL bring-up (ncclCommInitRank on every rank) may take before the run continues RCCL-free; "
                    "a calibration leg gets 1.5 x this")
    ap.add_argument("--no-exchange-forms", action="store_true", help="N>1: skip the legs that time the two exchange forms the timed region did not use (tests that do not look at them)")
    ap.add_argument("--pipeline", type=int, default=0, help="renderers in flight of the `throughput` measurement (each with its own HIP stream and, for N>1, its own "
                    "RCCL communicator); default 3 on one GPU, 4 for tile shards")
    ap.add_argument("--root-weight", type=int, default=0, help="N>1: tile-ownership weight of rank 0 against 8 for every other rank (lpt_renderer_set_shard_weighted): "
                    "rank 0 also unpacks, resolves and reads back the frame, so it gets fewer tiles; 0 = calibrate in the warm-up, 8 = equal shares")
    ap.add_argument("--opt", action="append", default=[], metavar="NAME=VALUE", help="experiments: lpt_renderer_set_option on every renderer (loupiote_amd._abi.OPTIONS: the five lpt_option values and the "
                    "LPT_OPT_EXPERIMENT knobs, by name: path_rays, coop_rays, tail_lanes, packet_primary, wavefront_rays; pipe_rays, refill, trace_waves_per_cu, ...); every value gives the same frame")
    ap.add_argument("--gltf", action="append", default=[], metavar="PATH", help="render the supplied glTF / GLB asset(s) — repeat the flag to load several into one scene, as the reference's "
                    "standalone does with DamagedHelmet.glb + sponza3.glb (crates/standalone/src/lib.rs:107-126) — instead of the synthetic stand-in: same span, same line, `data: real`")
    ap.add_argument("--probe", default=None, metavar="PATH.hdr", help="--gltf: a Radiance .hdr environment (the reference loads uffizi-large.hdr, lib.rs:109); default: constant grey")
    ap.add_argument("--camera", default=None, metavar="ox,oy,oz,dx,dy,dz", help="--gltf: camera origin and direction (default: the Cornell fixture's, (0,0.6,13.5) looking down -z)")
    ap.add_argument("--sort", type=int, default=0, help="experiments: lpt_renderer_set_sort_queues(flag) on every renderer (1 | 2: outgoing queues by octant)")
    ap.add_argument("--no-shard-emulation", action="store_true", help="skip the shard_emulation leg (rank 0's 1/2, 1/4, 1/8 tile shard of the frame on this GPU)")
    ap.add_argument("--blit-mode", choices=["pathtrace", "temporal", "denoised"], default="pathtrace",
                    help="BlitMode of every renderer (renderer.rs:160-167).  temporal / denoised: every raytrace() is a frame of the ASVGF pipeline (BASELINE config 5's form) — "
                         "on N>1 each call is followed by its exchange of the filter inputs and rank 0 filters; the tile weight is calibrated with the filter in the frame")
    ap.add_argument("--group-bracket", action="store_true", help="tests: every exchange inside lpt_comm_group_begin / lpt_comm_group_end (the form a host driving several GPUs from one thread uses)")
    ap.add_argument("--eager", action="store_true", help="experiments: every raytrace() launches at once (lpt_renderer_set_max_fused(1), the round-2 behaviour)")
    ap.add_argument("--max-fused", type=int, default=0, help="experiments: lpt_renderer_set_max_fused(n) on the timed renderer (0 = the library's default)")
    ap.add_argument("--lanes", type=int, default=0, help="experiments: wavefront lanes of the timed renderer (0 = the library's default)")
    ap.add_argument("--pageable", action="store_true", help="experiments: read_radiance() into pageable host memory")
    ap.add_argument("--oversubscribe", action="store_true", help="tests only: ranks may share GPUs (rank -> device rank %% device count); real RCCL refuses that, "
                    "so it needs LPT_RCCL_LIBRARY = the test stand-in (tests/tools/fake_rccl.c)")
    ap.add_argument("--spawn-dry-run", action="store_true", help="start the ranks and rendezvous over gloo only: no GPU is touched (CPU test of the launcher)")
    ap.add_argument("--spawn-timeout", type=float, default=1500.0, help="seconds the self-started ranks may take")
    return ap.parse_args(argv)


# ------------------------------------------------------------------------------------------------ launcher
def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def spawn_ranks(args):
    """`python bench.py --gpus N` without a launcher: start N ranks of this script, relay rank 0's JSON line.
    Runs before anything initialises a GPU (torch.cuda.device_count() does not, on this image); never exec()s."""
    n = args.gpus
    if not args.spawn_dry_run:
        import torch
        have = torch.cuda.device_count()
        if have < n and not (args.oversubscribe and have >= 1):
            sys.stderr.write("bench.py --gpus %d: only %d GPU(s) visible on this node — cannot start %d ranks (one process per GPU)\n" % (n, have, n))
            return 3
    env_base = dict(os.environ)
    env_base.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # the pool's host driver only supports dmabuf IPC: without it RCCL fails across processes (hipIpcGetMemHandle)
    env_base.update({"WORLD_SIZE": str(n), "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(free_port()), "LOCAL_WORLD_SIZE": str(n)})
    procs, logs = [], []
    tmp = tempfile.mkdtemp(prefix="lpt_bench_")
    for rank in range(n):
        env = dict(env_base)
        env.update({"RANK": str(rank), "LOCAL_RANK": str(rank)})
        err = open(os.path.join(tmp, "rank%d.err" % rank), "w+")
        out = subprocess.PIPE if rank == 0 else open(os.path.join(tmp, "rank%d.out" % rank), "w+")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env, stdout=out, stderr=err, text=True))
        logs.append(err)

    def tail(rank, nbytes=4000):
        logs[rank].flush()
        logs[rank].seek(0)
        return logs[rank].read()[-nbytes:]

    def stop_all():
        for p in procs:
            if p.poll() is None:
                p.terminate()
        t0 = time.time()
        while any(p.poll() is None for p in procs) and time.time() - t0 < 10:
            time.sleep(0.1)
        for p in procs:
            if p.poll() is None:
                p.kill()

    # rank 0's stdout is read to the end by communicate() in a thread-free way: poll the others meanwhile
    import threading
    box = {}
    reader = threading.Thread(target=lambda: box.setdefault("out", procs[0].stdout.read()), daemon=True)
    reader.start()
    t0 = time.time()
    failed = None
    while True:
        codes = [p.poll() for p in procs]
        bad = [i for i, c in enumerate(codes) if c not in (None, 0)]
        if bad:
            failed = bad[0]
            break
        if all(c == 0 for c in codes):
            break
        if time.time() - t0 > args.spawn_timeout:
            failed = -1
            break
        time.sleep(0.2)
    if failed is not None:
        stop_all()
        if failed < 0:
            sys.stderr.write("bench.py --gpus %d: the ranks did not finish within %.0f s; stopped them\n" % (n, args.spawn_timeout))
            for i in range(n):
                sys.stderr.write("---- rank %d stderr (tail) ----\n%s\n" % (i, tail(i, 1500)))
            return 4
        sys.stderr.write("bench.py --gpus %d: rank %d exited with code %s\n---- rank %d stderr (tail) ----\n%s\n" % (n, failed, procs[failed].returncode, failed, tail(failed)))
        return procs[failed].returncode or 1
    reader.join(timeout=10)
    lines = [l for l in (box.get("out") or "").splitlines() if l.startswith("{")]
    rc = 0
    if not lines:
        sys.stderr.write("bench.py --gpus %d: rank 0 printed no JSON line\n---- rank 0 stderr (tail) ----\n%s\n" % (n, tail(0)))
        rc = 5
    else:
        sys.stdout.write(lines[-1] + "\n")
        sys.stdout.flush()
    for f in logs:
        f.close()
    shutil.rmtree(tmp, ignore_errors=True)   # the ranks' logs: only of interest when something failed (reported above)
    return rc


def dry_run(args):
    """the launcher's CPU test: the ranks meet over gloo and rank 0 reports who came; no GPU, no library"""
    import torch
    import torch.distributed as dist
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    if os.environ.get("LPT_BENCH_TEST_DIE_RANK") == str(rank):   # launcher test: a rank that dies before the rendezvous
        sys.stderr.write("rank %d: LPT_BENCH_TEST_DIE_RANK set, exiting with code 7\n" % rank)
        return 7
    dist.init_process_group("gloo", rank=rank, world_size=world)
    t = torch.zeros(world, dtype=torch.int64)
    t[rank] = rank + 1
    dist.all_reduce(t)
    dist.barrier()
    dist.destroy_process_group()
    if rank == 0:
        print(json.dumps({"dry_run": True, "n_gpus": world, "ranks_seen": [int(x) - 1 for x in t.tolist()], "master": os.environ.get("MASTER_ADDR")}), flush=True)
    return 0


# ------------------------------------------------------------------------------------------------ helpers
def effective_cpus():
    """host threads this process may actually run on: the cgroup CPU quota when there is one (the GPU boxes expose 256
    hardware threads but grant 16 CPUs of time: oversubscribing them makes the oracle 40 % slower), else the affinity mask"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, q // per))
        except Exception:
            pass
    return n


def cpu_baseline(desc, view, threads, T):
    """The CPU oracle ("port") timed on the host cores on a bounded sample of the same workload:
    whole 1920x1080 frames of 1 spp each (the same seeds the GPU frame uses for its 1st, 2nd ... sample),
    as many as fit in ~12 s of wall time."""
    from oracle import orc  # checker only: the baseline leg, never the product path
    from oracle import harness
    flags = orc.use_native_build()   # -O3 -march=native, built here on the host it is timed on (this leg is the last thing the process does with the oracle)
    if "gltf" in desc:     # --gltf: the oracle's own loader reads the same bytes
        from oracle import gltf_oracle as G
        s = G.Scene()
        for blob in desc["gltf"]:
            G.load_gltf(blob, s)
        s.lights[0] = T.cornell_light()[0]
    else:
        s = harness.to_oracle(desc)
    sc = orc.OracleScene.from_scene(s, probe=desc["probe"])
    rays, secs, frames = 0, 0.0, 0
    while frames < 64 and secs < 12.0:   # whole 1-spp frames until ~12 s of wall time have been spent
        t0 = time.perf_counter()
        _, cnt = sc.render(WIDTH, HEIGHT, view, T.VFOV, DEPTH, frames=1, seed_counter=frames * DEPTH, threads=threads, want_counters=True)
        secs += time.perf_counter() - t0
        rays += cnt.closest + cnt.shadow
        frames += 1
    cpu = ""
    try:
        cpu = [l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")][0]
    except Exception:
        pass
    return {"value": rays / secs / 1e6, "unit": "Mrays/s", "cores": threads, "threads": threads, "flags": "gcc " + flags, "kind": "port", "nproc": os.cpu_count(), "cpu": cpu, "cpu_quota": "threads = the cgroup CPU quota (cpu.max) when there is one",
            "per_thread": rays / secs / 1e6 / max(threads, 1),
            "sample": "oracle/lpt_oracle.c (scalar C, persistent pthread pool, SAH BVH2, 16x16 tiles): %d whole 1-spp frames of the same 1920x1080 depth-8 "
                      "workload with the seeds of the GPU frame's 1st, 2nd ... sample (%.1f Mrays in %.1f s); a reported baseline, not a target" % (frames, rays / 1e6, secs)}


def baseline_metric():
    """the metric string exactly as BASELINE.json spells it"""
    try:
        return json.load(open(os.path.join(ROOT, "BASELINE.json")))["metric"]
    except Exception:
        return "Mrays/s (+ ms/frame) at 1920×1080, 4 spp, Sponza; 1/2/4/8 GPU"


def load_profile_json(name):
    path = os.path.join(ROOT, "profiles", name)
    if os.path.exists(path):
        try:
            return json.load(open(path))
        except Exception:
            return None
    return None


def kernel_source_hash():
    """sha256 (16 hex digits) over the device code: what profiles/limits.json and traffic.json record of the tree they were measured on (tools/limits_from_pmc.py)"""
    import hashlib
    h = hashlib.sha256()
    for f in ("kernels.h", "device_math.h"):
        try:
            h.update(open(os.path.join(ROOT, "loupiote_amd", "csrc", f), "rb").read())
        except OSError:
            pass
    return h.hexdigest()[:16]


def fixed_traversal_counts():
    """the per-unit figure of the roofline is FIXED per config — mean nodes visited / triangles tested per ray of config 4, committed with the
    fixtures (tests/golden/cfg4_traversal_counts.json: the counts of the kernel variant that fetches least, the two-round-trip step) — so that `roofline.frac`
    compares across builds: a build that fetches more per ray does not score higher.  None when the file is missing."""
    try:
        return json.load(open(os.path.join(ROOT, "tests", "golden", "cfg4_traversal_counts.json")))
    except Exception:
        return None


def from_profiles_object(avg_launch_ms):
    """What the bench line REPLAYS from committed profile summaries (profiles/traffic.json, profiles/limits.json: rocprofv3 --pmc passes of an
    earlier run of the same command, tools/profile.sh) — kept under ONE key so that a reader of the line cannot take it for something
    this run observed.  Returns (object, fabric bytes per k_trace launch or None)."""
    traffic_j = load_profile_json("traffic.json") or {}
    limits_j = load_profile_json("limits.json")
    traffic = traffic_j.get("k_trace_bytes_per_launch")
    binding = None
    if limits_j and "valu_issue" in limits_j and "gather_path" in limits_j:
        # the limit the kernel sits closest to.  Round 6's clean ceiling probes (profiles/r06_experiments_ab.txt A): +116 VALU instructions per node visit = +14.3 % of
        # the launch, one more 16-byte load per visit = +9.2 % — the launch follows the VALU issue rate first (0.97-0.98 of the microbenchmark ceiling of its instruction
        # mix) and the CU's vector-memory path second; both are reported, the larger fraction names the limit
        vf, gf = limits_j["valu_issue"]["frac"], limits_j["gather_path"]["frac_of_9.7"]
        binding = {"name": "valu_issue" if vf >= gf else "vector_memory_path", "frac": max(vf, gf),
                   "valu_issue_frac": vf, "vector_memory_path_frac_range": [limits_j["gather_path"]["frac_of_13.5"], gf],
                   "lane_efficiency": (limits_j.get("lane_efficiency") or {}).get("k_trace"),
                   "note": "the kernel's algorithmic bytes come from L1 / L2 / Infinity Cache, not from the HBM `frac` is quoted against.  Its launch time follows the VALU "
                           "instructions of a node visit (243 per visit, 0.12 % of the launch each: measured with clean A/B probes in round 6, DESIGN 5.1) issued at the "
                           "ceiling of their mix of full-rate and half-rate instructions (tools/microbench/f16_rate.hip), and second the 16-byte loads a visit pulls through "
                           "the CU's vector-memory path, quoted against the 9.7-13.5 TB/s a fully divergent dwordx4 gather reaches in the microbenchmark; at a lane "
                           "efficiency of 0.6 (divergence: SQ_THREAD_CYCLES_VALU / 64 SQ_INSTS_VALU)"}
    here = kernel_source_hash()
    measured_on = (limits_j or {}).get("kernel_source_hash") or traffic_j.get("kernel_source_hash")
    obj = {"what": "replayed from committed rocprofv3 --pmc summaries of an earlier run of this command — NOT measured in this run",
           "source": {"traffic": traffic_j.get("source"), "limits": (limits_j or {}).get("source")},
           # the device code the counters were measured on against the device code of this tree: a changed kernel with un-refreshed profiles says so
           "kernel_source_hash": {"profiles": measured_on, "this_tree": here}, "stale": measured_on != here,
           "traffic": traffic, "traffic_unit": "fabric bytes (FETCH_SIZE x 2 + WRITE_SIZE) per k_trace launch",
           "limits": limits_j, "binding_limit": binding}
    return obj, traffic


def roofline_fractions(achieved_gbps, traffic_bytes, avg_launch_ms):
    """frac = algorithmic bytes / launch time / 8 TB/s (the contract's figure: served mostly by the caches); frac_fabric = the bytes that crossed
    the fabric per launch (PMC, replayed) / THIS run's launch time / 8 TB/s — the HBM-side utilisation, never to be quoted without the other."""
    frac = achieved_gbps / HBM_PEAK_GBS
    frac_fabric = (traffic_bytes / (avg_launch_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if (traffic_bytes and avg_launch_ms > 0) else None
    return frac, frac_fabric


# ------------------------------------------------------------------------------------------------ one rank
def run(args):
    global WIDTH, HEIGHT, SPP
    import numpy as np
    import torch
    import torch.distributed as dist
    import loupiote_amd as lp

    if not os.path.exists(lp.LIB_PATH):  # the built library normally travels with the tree; a fresh checkout builds it (hipcc, ~1 min)
        if int(os.environ.get("LOCAL_RANK", "0")) == 0:
            from loupiote_amd import build as _build
            _build.build()
        else:  # one builder per node; the other ranks wait for the file
            _t0 = time.time()
            while not os.path.exists(lp.LIB_PATH) and time.time() - _t0 < 900:
                time.sleep(1.0)
            time.sleep(2.0)
    from loupiote_amd import scenes, testing as T

    WIDTH, HEIGHT, SPP = args.width, args.height, args.spp
    FPS = max(1, args.frames_per_step)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if args.oversubscribe:
        local_rank = local_rank % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local_rank)
    use_dist = world > 1 or args.force_dist
    if use_dist:
        if "MASTER_ADDR" not in os.environ:
            os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", "29517"
        dist.init_process_group("gloo", rank=rank, world_size=world)  # control plane only (id rendezvous, barriers, max over ranks)

    from loupiote_amd import api as lp_api
    for kv in args.opt:
        k, _, v = kv.partition("=")
        lp_api.DEFAULT_OPTIONS[k] = int(v)
    dev = lp.Device(local_rank)
    P = args.pipeline if args.pipeline > 0 else (4 if (world > 1 or args.emulate_shard > 1) else 3)
    extras = not args.no_extras
    # frames in flight over several communicators has never run on more than one GPU: on N>1 it is opt-in, so that an
    # untested leg cannot take the headline measurement down with it
    import threading
    auto = args.exchange == "auto"
    if auto and not use_dist:
        args.exchange = "gather"   # one rank: nothing is exchanged
        auto = False
    tp_leg = extras and (world == 1 or (args.throughput and args.exchange != "host"))
    host_gather = args.exchange == "host" and world > 1
    comms = []
    rccl_error = None        # auto: why the run continued RCCL-free
    hard_exit = [False]      # a thread was abandoned inside RCCL: leave through os._exit (its teardown may never return)

    def in_watchdog(fn, seconds, what):
        """fn() in a thread of its own, joined for `seconds`; the ranks then agree (gloo) on whether EVERY rank finished it.  A thread that did not return is abandoned where
        it is — never killed, the process never restarted — and the caller goes on without what it was bringing up.  Returns (result, error)."""
        res = {"value": None, "error": None}

        def work():
            try:
                res["value"] = fn()
            except Exception as e:   # noqa: BLE001 - anything RCCL throws must not cost the line
                res["error"] = "%s: %s: %s" % (what, type(e).__name__, e)
        th = threading.Thread(target=work, daemon=True)
        th.start()
        th.join(seconds)
        if th.is_alive():
            res["error"] = "%s did not return within %d s on rank %d (abandoned in its thread)" % (what, int(seconds), rank)
            hard_exit[0] = True
        errs = [None] * world
        dist.all_gather_object(errs, res["error"])
        err = next((e for e in errs if e), None)
        return res["value"], err

    if use_dist and not host_gather:
        # one communicator for the timed renderer + one per pipelined renderer of the throughput measurement: RCCL serialises
        # the operations of ONE communicator, so frames in flight must not share one
        n_comms = 1 + (P if tp_leg else 0)
        if auto:
            ids, rccl_error = in_watchdog(lambda: [lp.Comm.unique_id() for _ in range(n_comms)] if rank == 0 else None, 0.5 * args.rccl_timeout, "ncclGetUniqueId")
            if rccl_error is None:
                box = [ids]
                dist.broadcast_object_list(box, src=0)
                got, rccl_error = in_watchdog(lambda: [lp.Comm(dev, uid, rank, world) for uid in box[0]], args.rccl_timeout, "ncclCommInitRank")
                if rccl_error is None:
                    comms = got
            if rccl_error is not None and args.blit_mode != "pathtrace":
                raise SystemExit("bench.py: RCCL did not come up (%s) and --blit-mode %s needs it (the filter's inputs travel by RCCL)" % (rccl_error, args.blit_mode))
            if rccl_error is not None:
                print("bench.py: RCCL did not come up (%s): continuing with the host-side gather" % rccl_error, file=sys.stderr)
                args.exchange, auto = "host", False
                host_gather = world > 1
                tp_leg = False
        else:
            box = [[lp.Comm.unique_id() for _ in range(n_comms)] if rank == 0 else None]
            dist.broadcast_object_list(box, src=0)
            for uid in box[0]:
                comms.append(lp.Comm(dev, uid, rank, world))   # ncclCommInitRank inside the library (RCCL over xGMI)
    xmode = lp.EXCHANGE_REDUCE if args.exchange == "reduce" else lp.EXCHANGE_GATHER_TILES
    # host-side gather: ONE frame in POSIX shared memory that every rank maps and page-locks, and its frame barrier — behind the C ABI
    # (lpt_host_frame_*: shm + hipHostRegister + progress words with pause-spinning, then futex).  Every N>1 run gets one: the three exchange
    # forms are timed in one run (exchange_forms below)
    shared = None
    if world > 1 or (host_gather and use_dist):
        box = [("/lpt_frame_%d_%d" % (os.getpid(), int(time.time() * 1e6))) if rank == 0 else None]
        dist.broadcast_object_list(box, src=0)
        if rank == 0:
            shared = lp.HostFrame.create(box[0], WIDTH, HEIGHT, world)
        dist.barrier()
        if rank != 0:
            shared = lp.HostFrame.attach(box[0], WIDTH, HEIGHT, world)

    def host_frame_barrier():
        """every rank has written its pixels of this frame: lpt_host_frame_barrier (the frame number is the HostFrame's own counter)"""
        shared.barrier(rank)
    if args.gltf:
        # a real asset: the reference's loader path — load_gltf per file into ONE scene (lib.rs:117-123), the light of the Cornell fixture, an .hdr
        # environment if one is given — through the same C ABI the synthetic scene takes
        scene = lp.Scene()
        blobs = []
        for pth in args.gltf:
            blobs.append(open(pth, "rb").read())
            lp.loaders.load_gltf(blobs[-1], scene)
        scene.set_light(0, T.cornell_light())
        cam = [float(x) for x in args.camera.split(",")] if args.camera else list(T.CORNELL_EYE) + list(T.CORNELL_DIR)
        if len(cam) != 6:
            raise SystemExit("bench.py: --camera takes ox,oy,oz,dx,dy,dz")
        cnt = scene.counts()
        desc = {"gltf": blobs, "images": [None] * max(int(cnt.images) - 1, 0), "materials": [None] * max(int(cnt.materials) - 1, 0),   # (element 0 of every array is the reference's dummy)
                "probe": lp.load_env(open(args.probe, "rb").read()) if args.probe else T.CORNELL_PROBE,
                "camera": {"origin": tuple(cam[:3]), "direction": tuple(cam[3:])}, "name": " + ".join(os.path.basename(pth) for pth in args.gltf)}
        tex_bytes = 0     # (decoded inside the library; accel.texture_bytes_resident has the resident figure)
    else:
        desc = scenes.synthetic_atrium(textures=not os.environ.get("LPT_BENCH_NOTEX"), texture_size=args.texture_size)
        tex_bytes = int(sum(im.size for im in desc["images"]))
        scene = scenes.to_product(desc)
    sg = lp.SceneGPU.new_from_scene(scene, dev, gpu_build=bool(os.environ.get("LPT_BENCH_GPU_BUILD")))
    probe = lp.ProbeGPU(dev, desc["probe"], desc["probe"].shape[1], desc["probe"].shape[0])
    view = T.look(desc["camera"]["origin"], desc["camera"]["direction"])

    weights = [None]   # tile-ownership weights of the ranks (None = equal shares), the same on every rank
    blit_mode = {"pathtrace": lp.BlitMode.Pahtrace, "temporal": lp.BlitMode.Temporal, "denoised": lp.BlitMode.DenoisedPathrace}[args.blit_mode]
    denoising = blit_mode != lp.BlitMode.Pahtrace

    def exchange(rr, form=None):
        if args.group_bracket:
            lp.Comm.group_begin()
        rr.exchange(xmode if form is None else (lp.EXCHANGE_REDUCE if form == "reduce" else lp.EXCHANGE_GATHER_TILES))
        if args.group_bracket:
            lp.Comm.group_end()

    def make_renderer(comm=None, lanes=None, shard=None, host_form=None, wts="default"):
        """host_form: None = this run's main form (args.exchange); True / False = a renderer for the host-side gather / for an RCCL exchange"""
        shard = shard or args.emulate_shard
        host_form = host_gather if host_form is None else host_form
        wts = weights[0] if wts == "default" else wts
        rr = lp.Renderer(dev, (WIDTH, HEIGHT))
        if lanes:
            rr.set_lanes(lanes)
        if args.eager:
            rr.set_max_fused(1)
        rr.downsample_factor = 1.0
        rr.resize(dev, sg, probe, (WIDTH, HEIGHT))
        rr.set_max_bounces(DEPTH)
        rr.set_vfov(T.VFOV)
        if blit_mode != lp.BlitMode.Pahtrace:
            rr.set_blit_mode(blit_mode)
        if args.sort:
            rr.set_sort_queues(args.sort)
        if comm is not None and not host_form:
            rr.set_comm(comm, wts)                 # = set_shard(rank, world, 32, 8, weights) + the binding
            rr.set_resources(dev, sg, probe)
        elif host_form:
            rr.set_shard(rank, world, 32, 8, weights=wts)
            rr.set_resources(dev, sg, probe)
        elif shard > 1:
            rr.set_shard(0, shard, 32, 8)
            rr.set_resources(dev, sg, probe)
        return rr

    def fence(renderers):
        if use_dist:
            dist.barrier()
        for rr in renderers:
            rr.synchronize()
        torch.cuda.synchronize()

    # ================================================================== value: the SURVEY §8d span on ONE renderer
    r = make_renderer(comms[0] if comms else None, lanes=args.lanes or None)
    if args.max_fused:
        r.set_max_fused(args.max_fused)
    dst = None if args.pageable else lp.pinned_array((HEIGHT, WIDTH, 4))   # page-locked read-back destination (lpt_host_alloc)
    last = {}

    def span_frame(rr=None, form=None):
        """one frame in the SURVEY 8d span form on renderer rr (default: the timed one) ended by exchange form `form` (default: this run's)"""
        rr = r if rr is None else rr
        form = args.exchange if form is None else form
        rccl_form = use_dist and form != "host"
        rr.reset_accumulation()
        rr.accumulate = True                     # app.rs:318
        for _ in range(SPP):
            rr.raytrace(view)                    # records; the four calls leave together at the next submission point
            if rccl_form and denoising:
                exchange(rr, form)               # a denoising call is a frame of its own: its filter inputs travel, rank 0 filters
        if rccl_form and not denoising:
            exchange(rr, form)                   # RCCL on the renderer's stream, behind the frame's kernels; rank 0 presents the frame
        if use_dist and form == "host" and shared is not None:
            rr.read_radiance_owned(shared.array)     # every rank: its own pixels into the one shared frame, over its own link
            host_frame_barrier()                     # the frame is complete in host memory: the end of the §8d span
            if rank == 0:
                last["img"] = shared.array
        elif rank == 0:
            last["img"] = rr.read_radiance(out=dst)  # blocking: the end of the §8d span
        else:
            rr.synchronize()

    # --exchange auto: 5 calibration frames of every form that came up (equal tile shares, fresh renderers, each leg under the watchdog), the fastest ends the timed frames
    exchange_auto = None
    if auto:
        exchange_auto = {"what": "ms per frame (max over ranks) of 5 frames after 2 warm-up frames per exchange form, equal tile shares; the fastest is the timed region's",
                         "calibration_ms_per_frame": {}, "errors": {}}
        for form in ("host", "gather", "reduce"):
            if form == "host" and (shared is None or denoising):   # (the denoising modes exchange the filter's inputs: an RCCL form)
                continue

            def leg(form=form):
                rr = make_renderer(None if form == "host" else comms[0], lanes=args.lanes or None, host_form=(form == "host"), wts=None)
                for _ in range(2):
                    span_frame(rr, form)
                fence([rr])
                tc = time.perf_counter()
                for _ in range(5):
                    span_frame(rr, form)
                fence([rr])
                dt = torch.tensor([time.perf_counter() - tc], dtype=torch.float64)
                dist.all_reduce(dt, op=dist.ReduceOp.MAX)
                rr.close()
                return float(dt.item()) / 5 * 1e3
            ms, err = in_watchdog(leg, 1.5 * args.rccl_timeout, "calibration of the %s form" % form)
            if err is None:
                exchange_auto["calibration_ms_per_frame"][form] = ms
            else:
                exchange_auto["errors"][form] = err
                if form != "host":      # an RCCL operation that hangs or fails: no further RCCL in this run
                    rccl_error = err
                    break
        cal = exchange_auto["calibration_ms_per_frame"]
        box = [min(cal, key=cal.get) if cal else "host"]
        dist.broadcast_object_list(box, src=0)
        args.exchange = box[0]
        exchange_auto["chosen"] = args.exchange
        auto = False
        host_gather = args.exchange == "host" and world > 1
        xmode = lp.EXCHANGE_REDUCE if args.exchange == "reduce" else lp.EXCHANGE_GATHER_TILES
        if host_gather or rccl_error is not None:
            tp_leg = False
        r.close()
        r = make_renderer(comms[0] if (comms and not host_gather) else None, lanes=args.lanes or None)
        if args.max_fused:
            r.set_max_fused(args.max_fused)
    for _ in range(args.warmup):
        for _ in range(FPS):
            span_frame()
    fence([r])
    # N>1: rank 0 does more per frame than the others (unpack, resolve, the read-back): give it fewer tiles, so that its frame
    # takes as long as theirs.  Calibrated here on equal shares: c0 = what a frame costs beyond the slowest rank's tracing.
    calib = None
    if world > 1 and not host_gather:        # (the host-side gather gives rank 0 no extra work: equal shares)
        if args.root_weight:
            w0 = max(0, min(8, args.root_weight))
        else:
            CAL = 10
            r.enable_timings(True)
            fence([r])
            tc = time.perf_counter()
            for _ in range(CAL):
                span_frame()
            fence([r])
            t_frame = (time.perf_counter() - tc) / CAL * 1e3
            tm = r.timings()
            r.enable_timings(False)
            # rank 0's extra = its frame beyond the mean tracing time of the ranks: unpack, resolve, read-back — and, in the denoising modes, the filter
            t_tr = sum(tm.get(k, (0.0, 0))[0] for k in ("ray generation", "primary intersection", "intersection", "shading", "shadow", "path", "accumulation")) / CAL
            tt_ = torch.tensor([t_tr], dtype=torch.float64)
            dist.all_reduce(tt_, op=dist.ReduceOp.SUM)
            t_tr_mean = float(tt_.item()) / world
            c0 = max(0.0, t_frame - t_tr_mean)
            share0 = 1.0 - c0 * (world - 1) / max(world * t_tr_mean, 1e-9)      # rank 0's share relative to an equal share
            w0 = max(0 if denoising else 1, min(8, int(round(8.0 * share0))))   # the denoising modes may leave rank 0 as a pure compositor / filter
            calib = {"frame_ms_equal_shares": t_frame, "trace_ms_mean": t_tr_mean, "rank0_extra_ms": c0, "share0": share0}
        box = [w0]
        dist.broadcast_object_list(box, src=0)                                   # every rank uses rank 0's figure
        w0 = int(box[0])
        # the weights of a communicator may sum to at most 64 (kMaxVirtual): beyond 8 ranks the others' weight shrinks with the world size
        # (and past 32 ranks — kMaxWorld for weighted shards — the shares stay equal)
        base = max(1, min(8, 64 // max(world, 1)))
        w0 = max(0 if denoising else 1, min(base, (w0 * base + 4) // 8))   # round half up (not Python's banker's rounding); Pathtrace keeps rank 0 tracing
        if w0 != base and world <= 32:
            weights[0] = [w0] + [base] * (world - 1)
            r.close()
            r = make_renderer(comms[0], lanes=args.lanes or None)
            if args.max_fused:
                r.set_max_fused(args.max_fused)
            for _ in range(max(1, args.warmup)):
                for _ in range(FPS):
                    span_frame()
            fence([r])
    r.reset_ray_counts()
    fence([r])
    frame_ms = []
    sub0 = r.submission_stats()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        for _ in range(FPS):
            tf = time.perf_counter()
            span_frame()
            frame_ms.append((time.perf_counter() - tf) * 1e3)
    fence([r])
    elapsed = time.perf_counter() - t0
    sub1 = r.submission_stats()
    slowest_frame = max(range(len(frame_ms)), key=lambda i: frame_ms[i]) if frame_ms else -1
    frame_ms.sort()
    c_ = r.ray_counts()
    closest_l, shadow_l, shaded_l, primary_l = c_.closest, c_.shadow, c_.shaded, c_.primary
    n_frames = args.steps * FPS
    # the per-stage breakdown (and the exchange time) comes from FPS more frames of the same loop with the library's event timing on:
    # two hipEventRecord per launch delay the second wavefront's launches by ~0.1 ms per frame, so the timed region runs without them
    r.enable_timings(True)
    fence([r])
    for _ in range(FPS):
        span_frame()
    fence([r])
    timings = r.timings()
    r.enable_timings(False)
    r.reset_ray_counts()
    stage_ms_ranks = None
    if use_dist:   # load imbalance of the interleaved tiles: every stage's per-frame time on the slowest and on the fastest rank
        mine_ms = {k: v[0] / FPS for k, v in timings.items()}
        allm = [None] * world
        dist.all_gather_object(allm, mine_ms)
        stage_ms_ranks = {k: {"max": max(m.get(k, 0.0) for m in allm), "min": min(m.get(k, 0.0) for m in allm), "rank_of_max": max(range(world), key=lambda q: allm[q].get(k, 0.0))}
                          for k in sorted(set().union(*[set(m) for m in allm]))}

    tl = torch.tensor([elapsed], dtype=torch.float64)
    rays = torch.tensor([closest_l, shadow_l, shaded_l], dtype=torch.float64)
    per_rank = [float(closest_l + shadow_l)]
    if use_dist:
        dist.all_reduce(tl, op=dist.ReduceOp.MAX)
        gathered = [torch.zeros(1, dtype=torch.float64) for _ in range(world)]
        dist.all_gather(gathered, torch.tensor([float(closest_l + shadow_l)], dtype=torch.float64))
        per_rank = [float(g.item()) for g in gathered]
        dist.all_reduce(rays, op=dist.ReduceOp.SUM)
    elapsed = float(tl.item())
    closest, shadow, shaded = [float(x) for x in rays.tolist()]

    # the presented frame of the last timed step: every pixel of the frame with its 4 samples, on rank 0
    frame_ok = None
    checksum = None
    if rank == 0:
        img = last["img"]
        frame_ok = bool(np.all(img[..., 3] == 1.0) and np.all(np.isfinite(img)) and float(img[..., :3].mean()) > 0.0)
        checksum = float(np.float64(img[..., :3].sum()))
    if host_gather and use_dist:
        # `img` IS the shared frame: no rank may start the next frame (the stats frame below writes its pixels into it) before rank 0 has read this one
        dist.barrier()

    # ---- roofline of the dominant kernel (k_trace), this rank: algorithmic bytes per launch
    # = closest rays * (32 B ray read + 16 B hit write + N*80 B nodes + T*48 B triangles)
    # + shadow rays * (32 B ray read + 4 B + Ns*80 B + Ts*48 B), with N, T, Ns, Ts (mean nodes visited /
    # triangles tested per ray) measured by the stats variant of the same kernel on the same frame,
    # outside the timed region.
    r.enable_stats(True)
    r.reset_ray_counts()
    span_frame()
    fence([r])
    st = r.ray_counts()
    r_occ_cell = r.get_option("occ_cell_milli")
    r.enable_stats(False)
    # bounce 0 is traced by packet traversal (k_trace_packet: one tree walk per 64 coherent rays) — another kernel, with its own
    # counters; `nodes` / `tris` are k_trace's, per closest-hit ray of the bounces it traces
    kt_closest = st.closest - st.primary
    n_bar = st.nodes / max(kt_closest, 1)
    t_bar = st.tris / max(kt_closest, 1)
    packets_stat = max(st.primary / 64.0, 1.0)
    pk_nodes, pk_tris = st.packet_nodes / packets_stat, st.packet_tris / packets_stat
    ns_bar = st.shadow_nodes / max(st.shadow, 1)
    ts_bar = st.shadow_tris / max(st.shadow, 1)
    accel = sg.stats()
    b_ray = 32.0 + 16.0 + n_bar * accel.node_bytes + t_bar * accel.tri_bytes
    b_sh = 32.0 + 4.0 + ns_bar * accel.node_bytes + ts_bar * accel.tri_bytes

    def trace_stage(tm, cl, sh):
        # stages "intersection" = k_trace launches that carry closest-hit rays, "shadow" = the last, shadow-only launch
        ms = tm.get("intersection", (0.0, 0))[0] + tm.get("shadow", (0.0, 0))[0]
        launches = tm.get("intersection", (0.0, 0))[1] + tm.get("shadow", (0.0, 0))[1]
        avg = ms / max(launches, 1)
        byts = (cl * b_ray + sh * b_sh) / max(launches, 1)
        return avg, launches, byts, (byts / (avg * 1e-3) / 1e9 if avg > 0 else 0.0)

    # In the timed region the 4 samples of a frame leave as two wavefronts (half the tile rows each) on the renderer's two lanes and overlap: a launch there
    # shares the chip.  The kernel figure (roofline.frac) is taken from SOLO_FRAMES frames issued as ONE wavefront each
    # (lpt_renderer_set_max_fused(spp)), one frame at a time: un-overlapped launches, HIP events on the stream they run on.
    SOLO_FRAMES = 3
    o_avg, o_launches, o_bytes, o_achieved = trace_stage(timings, (closest_l - primary_l) / n_frames * FPS, shadow_l / n_frames * FPS)
    r.set_max_fused(max(SPP, 1))
    span_frame()                       # the lanes' ray buffers grow to the batch size on first use
    fence([r])
    r.reset_ray_counts()
    r.enable_timings(True)
    for _ in range(SOLO_FRAMES):
        span_frame()
    fence([r])
    solo_t = r.timings()
    r.enable_timings(False)
    sc_ = r.ray_counts()
    r.set_max_fused(args.max_fused)
    s_avg, s_launches, s_bytes, s_achieved = trace_stage(solo_t, sc_.closest - sc_.primary, sc_.shadow)
    rays_per_launch = (sc_.closest - sc_.primary + sc_.shadow) / max(s_launches, 1)
    # the roofline's per-ray bytes are a FIXED figure per config (committed counts), so that `frac` compares across builds; what THIS build's kernel
    # really fetched (its own stats variant, above) is `frac_fetched` — a build that fetches more per ray scores higher there, not in `frac`
    fixed = fixed_traversal_counts()
    if fixed and WIDTH == 1920 and HEIGHT == 1080 and SPP == 4:
        # the committed figure as a whole — SURVEY 8d's S_node = 80 B included: round 6's 64-byte node fetches LESS than the contract's algorithmic bytes, and a
        # build that moves fewer bytes for the same rays must not score lower for it (what this build's kernel really fetched: frac_fetched, with ITS node size)
        fb_ray = float(fixed.get("bytes_per_ray", 32.0 + 16.0 + fixed["nodes_per_ray"] * 80.0 + fixed["tris_per_ray"] * 48.0))
        fb_sh = float(fixed.get("bytes_per_shadow_ray", 32.0 + 4.0 + fixed["shadow_nodes_per_ray"] * 80.0 + fixed["shadow_tris_per_ray"] * 48.0))
        fixed_bytes = ((sc_.closest - sc_.primary) * fb_ray + sc_.shadow * fb_sh) / max(s_launches, 1)
        fixed_achieved = fixed_bytes / (s_avg * 1e-3) / 1e9 if s_avg > 0 else 0.0
    else:   # another workload than config 4 (experiments), or no committed counts: the fetched figure is all there is
        fixed, fb_ray, fb_sh, fixed_bytes, fixed_achieved = None, b_ray, b_sh, s_bytes, s_achieved
    pk_ms, pk_launches = solo_t.get("primary intersection", (0.0, 0))
    packet_j = None
    if pk_launches:
        pk_avg = pk_ms / pk_launches
        pk_bytes = 32.0 * 64 + 16.0 * 64 + pk_nodes * accel.node_bytes + pk_tris * accel.tri_bytes      # per packet: 64 rays read, 64 hits written, the nodes / triangles once
        packet_j = {"kernel": "k_trace_packet", "what": "bounce 0: one tree walk per 64 coherent primary rays (the four samples of a 4x4-pixel patch), node and triangle fetches by scalar loads",
                    "avg_launch_ms": pk_avg, "launches": pk_launches, "rays_per_launch": sc_.primary / pk_launches, "Mrays_per_s": sc_.primary / pk_launches / (pk_avg * 1e-3) / 1e6,
                    "Grays_per_s": sc_.primary / pk_launches / (pk_avg * 1e-3) / 1e9,
                    "nodes_per_packet": pk_nodes, "tris_per_packet": pk_tris, "bytes_per_packet": pk_bytes,
                    "achieved_GBps": sc_.primary / 64.0 / pk_launches * pk_bytes / (pk_avg * 1e-3) / 1e9}
    exchange_ms = timings.get("exchange", (0.0, 0))

    # ---- the read-back alone: k_resolve + 33 MB device -> host of an already finished frame (depends on the box's PCIe link and host)
    readback = None
    if rank == 0:
        rb = []
        for _ in range(8):
            r.synchronize()
            t1 = time.perf_counter()
            r.read_radiance(out=dst)
            rb.append((time.perf_counter() - t1) * 1e3)
        rb.sort()
        readback = {"median_ms": rb[len(rb) // 2], "min_ms": rb[0], "bytes": WIDTH * HEIGHT * 16, "GBps": WIDTH * HEIGHT * 16 / (rb[len(rb) // 2] * 1e-3) / 1e9,
                    "destination": "pageable" if args.pageable else "page-locked (lpt_host_alloc)"}

    # ---- latency: one frame alone, host call to completion, no read-back
    latency = None
    if extras:
        lat = []
        for _ in range(7):
            fence([r])
            t1 = time.perf_counter()
            r.reset_accumulation()
            r.accumulate = True
            for _ in range(SPP):
                r.raytrace(view)
            if comms:
                exchange(r)
            r.synchronize()
            lat.append((time.perf_counter() - t1) * 1e3)
        lat.sort()
        latency = {"min": lat[0], "median": lat[len(lat) // 2], "frames": len(lat),
                   "what": "one frame alone: reset_accumulation + 4 x raytrace(view)%s + stream synchronize (no read-back)" % (" + exchange" if comms else "")}
        if readback is not None:
            # what the span pays for its read-back: the read that ends a frame submits it and copies every wavefront's rows as soon as they are
            # final (include/lpt.h lpt_renderer_read_radiance), so part of `median_ms` hides under the wavefronts that are still tracing
            readback["exposed_in_span_ms"] = elapsed / n_frames * 1e3 - latency["median"]
    from loupiote_amd import _abi as lp_abi
    lib_options = {name: int(r.get_option(name)) for name in sorted(lp_abi.OPTIONS)}   # what the library chose / was told: the line says which pipeline it timed
    r.close()

    # ================================================================== strong scaling of ONE frame, emulated: rank 0's 1/N tile shard on this GPU
    # What the tracing of a frame costs a rank of an N-GPU job (interleaved 32x8 tiles, replicated scene), in the span form —
    # reset; 4 x raytrace; read_radiance of the whole frame buffer, one frame at a time — measured here because the driver has no
    # multi-GPU node to run `--gpus N` on.  No exchange is in it: an upper bound of the speed-up N GPUs can give one frame.
    shard_emulation = None
    if extras and world == 1 and not args.emulate_shard and not args.no_shard_emulation:
        shard_emulation = {"what": "ms per frame of rank 0's 1/N tile shard (32x8 tiles, tile id mod N) rendered alone on this GPU in the span form (reset_accumulation; %d x raytrace; "
                                   "read_radiance of the WHOLE frame buffer: what rank 0 of the RCCL-gather form pays), median of 12 frames after 4 warm-up frames; no exchange; '1' = the whole frame, the "
                                   "same way.  ms_per_frame_host_gather: the same frames ended by lpt_renderer_read_radiance_owned — only the rank's own pixels travel to the (shared) host "
                                   "frame, each GPU over its own PCIe link: the span of EVERY rank in the host-side-gather form (DESIGN 6), before the host barrier" % SPP}
        dst_owned = dst if dst is not None else lp.pinned_array((HEIGHT, WIDTH, 4))
        for n_sh in (1, 2, 4, 8):
            rr = make_renderer(lanes=args.lanes or None, shard=n_sh)
            ts = []
            for k in range(16):
                rr.synchronize()
                t1 = time.perf_counter()
                rr.reset_accumulation()
                rr.accumulate = True
                for _ in range(SPP):
                    rr.raytrace(view)
                rr.read_radiance(out=dst)
                ts.append((time.perf_counter() - t1) * 1e3)
            ts = sorted(ts[4:])
            th = []                                  # the same frames with the HOST-SIDE GATHER's read-back: only this rank's pixels travel
            for k in range(12):
                rr.synchronize()
                t1 = time.perf_counter()
                rr.reset_accumulation()
                rr.accumulate = True
                for _ in range(SPP):
                    rr.raytrace(view)
                rr.read_radiance_owned(dst_owned)
                th.append((time.perf_counter() - t1) * 1e3)
            th = sorted(th[2:])
            cc = rr.ray_counts()
            rr.enable_timings(True)
            rr.reset_accumulation()
            rr.accumulate = True
            for _ in range(SPP):
                rr.raytrace(view)
            rr.synchronize()
            stg = {k: v[0] for k, v in rr.timings().items() if v[1]}
            rr.close()
            shard_emulation[str(n_sh)] = {"ms_per_frame": ts[len(ts) // 2], "min_ms": ts[0], "ms_per_frame_host_gather": th[len(th) // 2],
                                          "rays_per_frame": (cc.closest + cc.shadow) / 28.0, "stage_ms": stg}
        for n_sh in (2, 4, 8):
            shard_emulation[str(n_sh)]["speedup_vs_1"] = shard_emulation["1"]["ms_per_frame"] / shard_emulation[str(n_sh)]["ms_per_frame"]
            shard_emulation[str(n_sh)]["speedup_vs_1_host_gather"] = shard_emulation["1"]["ms_per_frame"] / shard_emulation[str(n_sh)]["ms_per_frame_host_gather"]

    # ================================================================== throughput: P renderers in flight, batched samples, no read-back
    throughput = None
    if tp_leg:
        rs = [make_renderer(comms[1 + k] if comms else None, lanes=1) for k in range(P)]
        T_STEPS = max(2, min(args.steps, 6))
        no = [0]

        def tp_frame():
            rr = rs[no[0] % P]
            no[0] += 1
            rr.reset_accumulation()
            rr.accumulate = True
            rr.raytrace_n(view, SPP)      # == SPP x { raytrace(view); accumulate = true } as one wavefront, submitted at once
            if comms:
                exchange(rr)

        for _ in range(2 * FPS):
            tp_frame()
        fence(rs)
        for rr in rs:
            rr.reset_ray_counts()
            rr.enable_timings(True)
        fence(rs)
        t1 = time.perf_counter()
        for _ in range(T_STEPS * FPS):
            tp_frame()
        fence(rs)
        dt = time.perf_counter() - t1
        tm = {}
        tp_rays = 0
        for rr in rs:
            for k, v in rr.timings().items():
                a, b = tm.get(k, (0.0, 0))
                tm[k] = (a + v[0], b + v[1])
            rr.enable_timings(False)
            cc = rr.ray_counts()
            tp_rays += cc.closest + cc.shadow
        tt = torch.tensor([dt], dtype=torch.float64)
        tr = torch.tensor([float(tp_rays)], dtype=torch.float64)
        if use_dist:
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dist.all_reduce(tr, op=dist.ReduceOp.SUM)
        o_avg, o_launches, o_bytes, o_achieved = trace_stage(tm, (closest_l - primary_l) / n_frames * T_STEPS * FPS, shadow_l / n_frames * T_STEPS * FPS)
        throughput = {"value": float(tr.item()) / float(tt.item()) / 1e6, "unit": "Mrays/s", "ms_per_frame": float(tt.item()) / (T_STEPS * FPS) * 1e3,
                      "frames": T_STEPS * FPS, "frames_in_flight": P, "communicators": P if comms else 0,
                      "what": "%d renderers take the frames in turn (own HIP streams%s); frame = reset_accumulation + raytrace_n(view, 4)%s; no read-back — "
                              "round 2's headline figure" % (P, ", own RCCL communicators" if comms else "", " + lpt_renderer_exchange" if comms else ""),
                      "k_trace_overlapped": {"avg_launch_ms": o_avg, "launches": o_launches, "frac": o_achieved / HBM_PEAK_GBS,
                                             "note": "HIP events around every k_trace launch while %d frames share the chip: a scheduling figure, not a kernel figure" % P},
                      "stage_ms_per_frame": {k: v[0] / (T_STEPS * FPS) for k, v in tm.items()}}
        for rr in rs:
            rr.close()

    host_gather_j = None
    if host_gather:
        host_gather_j = {"what": "no exchange on the GPUs: every rank wrote its owned pixels of the mean radiance into ONE shared-memory frame (lpt_renderer_read_radiance_owned), "
                                 "a host-side barrier on shared words completed it", "frame_complete_on_rank0": frame_ok, "per_rank_rays": per_rank, "ranks": world}
    rccl = None
    if rccl_error is not None:
        rccl = {"error": rccl_error, "what": "--exchange auto: RCCL did not come up or an RCCL form failed under the watchdog; the run continued with the forms that work"}
    elif comms and not host_gather:
        rk, nr = comms[0].info()
        rccl = {"rccl_nranks": nr, "rccl_rank": rk, "communicators_per_rank": len(comms),
                "exchange_frame_complete_on_rank0": frame_ok,
                "exchange_ms_per_frame_rank0": exchange_ms[0] / max(exchange_ms[1], 1), "exchanges_timed": exchange_ms[1],
                "exchange_ms_what": "HIP events on rank 0's renderer stream from the pack kernel to the end of the unpack: includes waiting for the slowest rank's tiles",
                "per_rank_rays": per_rank, "mode": args.exchange,
                "tile_weights": weights[0] or [1] * world, "tile_weight_calibration": calib,
                "tile_weights_what": "rank 0 unpacks, resolves and reads back every frame besides tracing: it owns fewer tiles (lpt_renderer_set_shard_weighted), "
                                     "so that its frame takes as long as the others' — the image does not depend on the weights"}

    out = None
    if rank == 0:
        profiles_j, fabric_bytes = from_profiles_object(s_avg)
        rf_frac, rf_frac_fabric = roofline_fractions(fixed_achieved, fabric_bytes, s_avg)
        rf_frac_fetched = s_achieved / HBM_PEAK_GBS
        out = {
            "metric": baseline_metric(),
            "value": (closest + shadow) / elapsed / 1e6,
            "unit": "Mrays/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "real" if args.gltf else "synthetic",
            "config": {"workload": ("--gltf %s" % desc["name"] if args.gltf else "synthetic_atrium(seed=2)") + " [%s%d triangles, %d textures = %.1f MB of texels, %d materials], %dx%d, %d spp, depth %d, "
                                   "camera %s; frame = the SURVEY 8d span on one renderer per GPU: reset_accumulation; %d x raytrace(view) "
                                   "(recorded; submitted by the read as 4-sample wavefronts over runs of tile rows)%s; read_radiance() into %s host memory on rank 0; step = %d frames"
                                   % ("" if args.gltf else "Sponza stand-in: ", accel.triangles, len(desc["images"]), (accel.texture_bytes_resident if args.gltf else tex_bytes) / 1e6, len(desc["materials"]),
                                      WIDTH, HEIGHT, SPP, DEPTH, "%r->%r" % (tuple(desc["camera"]["origin"]), tuple(desc["camera"]["direction"])) if args.gltf else "(-10,1,0)->(1,0.35,0)", SPP,
                                      "; lpt_renderer_exchange(%s)" % args.exchange if use_dist else "", "pageable" if args.pageable else "page-locked", FPS),
                       "texture_bytes": tex_bytes, "textures": len(desc["images"]), "materials": len(desc["materials"]),
                       "texture_bytes_resident": int(accel.texture_bytes_resident), "texture_pairs": int(accel.texture_pairs),
                       "frames_per_step": FPS, "frames_timed": n_frames, "timed_region_s": elapsed, "submission": "eager" if args.eager else "record-then-submit",
                       "raytrace_calls_per_frame": (sub1[0] - sub0[0]) / max(args.steps * FPS, 1), "wavefronts_per_frame": (sub1[1] - sub0[1]) / max(args.steps * FPS, 1),
                       "tiles": "32x8 interleaved, tile_id mod N", "exchange": ((args.exchange + (" (host-side gather: lpt_host_frame_*, no collective)" if args.exchange == "host" else " (native RCCL, lpt_renderer_exchange)")
                                     + (" — picked by --exchange auto" if exchange_auto or rccl_error else "")) if use_dist else "none"),
                       "rays_per_frame": (closest + shadow) / n_frames, "rays_per_step": (closest + shadow) / args.steps,
                       "closest_rays": closest, "shadow_rays": shadow, "shaded_hits": shaded, "frame_complete": frame_ok, "frame_checksum": checksum,
                       "library_options": lib_options},
            "ms_per_frame": elapsed / n_frames * 1e3,
            "frame_ms_percentiles": {"min": frame_ms[0], "p10": frame_ms[len(frame_ms) // 10], "median": frame_ms[len(frame_ms) // 2],
                                     "p90": frame_ms[(9 * len(frame_ms)) // 10], "max": frame_ms[-1], "index_of_max": slowest_frame,
                                     "what": "host wall time of each timed frame on rank 0 (reset ... read_radiance returns)"},
            "throughput": throughput,
            "latency_ms": latency,
            "readback": readback,
            "shard_emulation": shard_emulation,
            "rccl": rccl,
            "host_gather": host_gather_j,
            "exchange_forms": None,      # filled in below: the legs run after the line is assembled (a watchdog prints it if they hang)
            "exchange_auto": exchange_auto,
            "stage_ms_per_rank": stage_ms_ranks,
            "roofline": {"bound": "hbm", "kernel": "k_trace", "achieved": fixed_achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": rf_frac,
                         # `frac` / `achieved`: the FIXED algorithmic bytes per ray of config 4 (tests/golden/cfg4_traversal_counts.json) x the rays a launch carries —
                         # comparable across builds.  `frac_fetched`: the bytes THIS build's kernel fetched per ray (its own stats variant): last round's `frac`.
                         "frac_fetched": rf_frac_fetched, "achieved_fetched": s_achieved, "bytes_per_launch_fetched": s_bytes,
                         "fixed_counts": fixed, "bytes_per_ray_fixed": fb_ray, "bytes_per_shadow_ray_fixed": fb_sh,
                         # the comparable speed figure: rays per second of the un-overlapped launch
                         "Grays_per_s": rays_per_launch / (s_avg * 1e-3) / 1e9 if s_avg > 0 else None,
                         # the same launches against the bytes that actually crossed the fabric (replayed counter figure / this run's launch time): the
                         # HBM-side utilisation.  `frac` says the caches serve the algorithmic bytes as fast as HBM could; this says how busy HBM is.
                         "frac_fabric": rf_frac_fabric,
                         "traffic": fabric_bytes, "traffic_is": "from_profiles (replayed, see roofline.from_profiles.source), not measured in this run",
                         "avg_launch_ms": s_avg, "launches": s_launches, "bytes_per_launch": fixed_bytes,
                         "basis": "un-overlapped launches: %d frames issued as one 4-sample wavefront each (lpt_renderer_set_max_fused(4)), one frame at a time, HIP events on the "
                                  "stream the kernel runs on around every k_trace launch — the duration rocprofv3's kernel trace reports for `bench.py --max-fused 4 --lanes 1` "
                                  "(profiles/*_solo_kernel_stats.csv)" % SOLO_FRAMES,
                         "timed_region": {"achieved": o_achieved, "frac": o_achieved / HBM_PEAK_GBS, "avg_launch_ms": o_avg, "launches": o_launches, "bytes_per_launch": o_bytes,
                                          "note": "the same over the timed region, where the two 4-sample half-frame wavefronts of a frame overlap on the renderer's lanes: a launch shares "
                                                  "the chip with the other wavefront's kernels — a scheduling figure, not a kernel figure"},
                         "region": {"achieved": ((closest_l - primary_l) * b_ray + shadow_l * b_sh) / elapsed / 1e9, "frac": ((closest_l - primary_l) * b_ray + shadow_l * b_sh) / elapsed / 1e9 / HBM_PEAK_GBS,
                                    "note": "all k_trace algorithmic bytes of the timed region / its wall time (which also contains k_shade, ray generation, accumulation, the read-back): a lower bound"},
                         "from_profiles": profiles_j,
                         "rays_per_launch": rays_per_launch, "bytes_per_ray": b_ray, "bytes_per_shadow_ray": b_sh, "nodes_per_ray": n_bar, "tris_per_ray": t_bar,
                         "shadow_nodes_per_ray": ns_bar, "shadow_tris_per_ray": ts_bar,
                         "occluder_cache_probe": {"what": "stats kernels only: what a table of the last occluding triangle per %.2f-unit cell of the shadow rays' origins would have answered "
                                                          "(found = an entry was there, hits = its triangle occludes the ray); no kernel uses such a cache" % (r_occ_cell * 1e-3),
                                                  "shadow_rays": st.shadow, "occluded": st.shadow_occluded, "found": st.occluder_cache_found, "hits": st.occluder_cache_hits,
                                                  "occluded_frac": st.shadow_occluded / max(st.shadow, 1), "hit_frac_of_shadow_rays": st.occluder_cache_hits / max(st.shadow, 1),
                                                  "hit_frac_of_occluded": st.occluder_cache_hits / max(st.shadow_occluded, 1)},
                         "wave": {"live_lanes_per_step": st.live_lanes / max(st.wave_steps, 1), "node_lanes_per_step": st.node_lanes / max(st.wave_steps, 1),
                                  "tri_lanes_per_step": st.tri_lanes / max(st.wave_steps, 1), "lane_slots_per_ray": 64.0 * st.wave_steps / max(kt_closest, 1)},
                         "packet": packet_j},
            "stage_ms_per_frame": {k: v[0] / FPS for k, v in timings.items()},
            "stage_ms_per_frame_solo": {k: v[0] / SOLO_FRAMES for k, v in solo_t.items()},
            "accel": {"triangles": accel.triangles, "nodes": accel.nodes, "node_bytes": accel.node_bytes,
                      "tri_bytes": accel.tri_bytes, "depth": accel.max_depth, "build_ms": accel.build_ms},
        }
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(desc, view, effective_cpus(), T)
        elif world > 1:
            out["cpu_baseline"] = None
    # The legs below run RCCL calls that have never executed with more than one rank anywhere (ncclReduce; ncclSend / ncclRecv when the main form was `host`): a hang
    # there must not cost the line.  A watchdog on every rank: after 180 s rank 0 prints the line it has (exchange_forms = the legs finished so far + the
    # timeout) and every rank leaves at once.
    wd_state = {"done": False}

    def wd_bail():
        if wd_state["done"]:
            return
        if rank == 0 and out is not None:
            out["exchange_forms"] = dict(exchange_forms or {}, error="timed out after 180 s: the remaining legs did not finish")
            try:
                import ctypes
                ctypes.CDLL(None).fflush(None)
            except Exception:
                pass
            sys.stdout.flush()
            print(json.dumps(out), flush=True)
        os._exit(3)      # the line is out, but a leg hung: launchers and CI must see that
    watchdog = threading.Timer(180.0, wd_bail)
    watchdog.daemon = True
    if use_dist and (world > 1 or args.force_dist):
        watchdog.start()
    # ================================================================== N>1: the three exchange forms in ONE run ("implement ncclReduce first, then the compact
    # variant, and report both"; the host-side gather is the third).  The timed region above used `--exchange`; the other two get 20 frames each here, on fresh
    # renderers (RCCL forms: the calibrated tile weights if the main form calibrated them, else equal shares; host form: equal shares), same span, max over ranks.
    exchange_forms = None
    lazy_comms = []
    if use_dist and (world > 1 or args.force_dist) and not args.emulate_shard and not denoising and not args.no_exchange_forms:
        main_form = args.exchange
        exchange_forms = {"what": "ms per frame and Mrays/s of the SAME frame ended by each exchange form, one process per GPU: gather = owned tiles by grouped ncclSend/ncclRecv "
                                  "+ read_radiance on rank 0; reduce = ncclReduce(sum) of the whole radiance buffer + read_radiance on rank 0; host = every rank writes its own pixels "
                                  "into one shared host frame (lpt_host_frame_*), no exchange on the GPUs.  `%s` is the timed region of this line, the others 20 frames after 3 warm-up "
                                  "frames.  frame_checksum = sum of the RGB of the FIRST frame of a fresh renderer ended by that form, on rank 0 (a renderer's seed counter never "
                                  "rewinds, so only frames of equal index compare) — the three must be equal" % main_form,
                          main_form: {"ms_per_frame": elapsed / n_frames * 1e3, "Mrays_s": (closest + shadow) / elapsed / 1e6, "frames": n_frames, "timed_region": True}}
        for form in ("gather", "reduce", "host"):
            try:
                if form == "host":
                    if shared is None:
                        raise RuntimeError("no shared host frame (one rank)")
                    rr = make_renderer(None, lanes=args.lanes or None, host_form=True, wts=None)
                else:
                    if rccl_error is not None:         # --exchange auto found RCCL unusable: not tried again
                        raise RuntimeError(rccl_error)
                    if not comms and not lazy_comms:   # the main form was `host`: RCCL comes up only now (and a failure here cannot take the headline down)
                        box = [lp.Comm.unique_id() if rank == 0 else None]
                        dist.broadcast_object_list(box, src=0)
                        lazy_comms.append(lp.Comm(dev, box[0], rank, world))
                    rr = make_renderer((comms or lazy_comms)[0], lanes=args.lanes or None, host_form=False)
                span_frame(rr, form)    # the first frame of a fresh renderer: the one whose checksum compares across the forms
                ck = float(np.float64(last["img"][..., :3].sum())) if rank == 0 else None
                dist.barrier()          # the host form's image IS the shared frame: read before anybody renders into it again
                if form == main_form:
                    exchange_forms[form]["frame_checksum"] = ck      # its times are the timed region's
                else:
                    for _ in range(2):
                        span_frame(rr, form)
                    rr.reset_ray_counts()
                    fence([rr])
                    t1 = time.perf_counter()
                    for _ in range(20):
                        span_frame(rr, form)
                    fence([rr])
                    dt = torch.tensor([time.perf_counter() - t1], dtype=torch.float64)
                    cc = rr.ray_counts()
                    ry = torch.tensor([float(cc.closest + cc.shadow)], dtype=torch.float64)
                    dist.all_reduce(dt, op=dist.ReduceOp.MAX)
                    dist.all_reduce(ry, op=dist.ReduceOp.SUM)
                    exchange_forms[form] = {"ms_per_frame": float(dt.item()) / 20 * 1e3, "Mrays_s": float(ry.item()) / float(dt.item()) / 1e6, "frame_checksum": ck, "frames": 20, "timed_region": False}
                rr.close()
            except Exception as e:   # noqa: BLE001 - an extra leg must not cost the line
                if form == main_form:
                    exchange_forms[form]["error"] = "%s: %s" % (type(e).__name__, e)
                else:
                    exchange_forms[form] = {"error": "%s: %s" % (type(e).__name__, e)}
        if rank == 0:
            cks = [v.get("frame_checksum") for k, v in exchange_forms.items() if isinstance(v, dict) and "frame_checksum" in v]
            exchange_forms["checksums_equal"] = bool(len(cks) >= 2 and all(c == cks[0] for c in cks))
            if not exchange_forms["checksums_equal"]:
                print("bench.py: the exchange forms do not end in the same frame: %r" % ({k: v for k, v in exchange_forms.items() if k != "what"},), file=sys.stderr)

    wd_state["done"] = True
    watchdog.cancel()
    if out is not None:
        out["exchange_forms"] = exchange_forms
    if use_dist:
        dist.barrier()
    if shared is not None:
        last.pop("img", None)
        shared.close()       # unregister + unmap; the creator (rank 0) unlinks
    if hard_exit[0]:
        # a thread sits in an RCCL call that never returned: a communicator's teardown may not return either.  The line is complete: print it and leave
        if out is not None:
            sys.stdout.flush()
            print(json.dumps(out), flush=True)
        os._exit(0)
    for c in comms + lazy_comms:
        c.close()
    if use_dist:
        dist.destroy_process_group()
    probe.close()
    sg.close()
    dev.close()
    if out is not None:
        # RCCL writes its version banner to stdout through C stdio: push that out first so that the JSON line is the
        # last thing on rank 0's stdout
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        sys.stdout.flush()
        print(json.dumps(out), flush=True)
    return 0


def main():
    args = parse_args()
    if args.gpus < 1:
        raise SystemExit("bench.py: --gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        return spawn_ranks(args)       # parent: never touches a GPU
    if args.spawn_dry_run:
        if "WORLD_SIZE" not in os.environ:
            os.environ.update({"RANK": "0", "WORLD_SIZE": "1", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(free_port())})
        return dry_run(args)
    return run(args)


if __name__ == "__main__":
    sys.exit(main())
