#!/usr/bin/env python3
"""bench.py — Mrays/s of the path-tracing hot path on N MI355X (one process per GPU).

Workload (BASELINE.json metric: "Mrays/s (+ ms/frame) at 1920x1080, 4 spp, Sponza"):
  synthetic_atrium(seed=2) — the Sponza STAND-IN (the real asset is absent) —
  1920x1080, 4 spp (= 4 raytrace() calls with accumulate), path depth 8, camera = the reference's
  start pose.  A FRAME = reset_accumulation(); accumulate=true; 4 x Renderer::raytrace(view) — issued as
  lpt_renderer_raytrace_n(view, 4), the bit-identical batched form; for N>1 followed by
  lpt_renderer_exchange (native RCCL inside the library: owned-tile gather to rank 0, or --exchange reduce).
  A STEP = FRAMES_PER_STEP (10) such frames, so that the driver's `--steps 20` times about 2 s instead of 0.2 s.
  Inputs (scene, BVH, probe, textures) are resident in HBM before the timed region.
  Consecutive frames rotate over three renderers with their own HIP streams (--pipeline 3: three frames in flight,
  triple buffering; four for the smaller wavefronts of a tile shard), so the tail of frame k (and its exchange) overlaps the heads of the next frames; every frame is
  still one complete frame.  GPU_MAX_HW_QUEUES is raised to 8 (ROCm default 4, of which the streams here got two):
  with fewer hardware queues than streams the frames serialise again.
  N>1: frames shard by interleaved 32x8 pixel tiles (tile id mod N), per-GPU work shrinks as
  N grows ("strong" scaling of one frame).  torch.distributed (gloo) is the control plane only — rendezvous of the
  128-byte RCCL id, barriers, the max over ranks; the data path is lpt_renderer_exchange.

value = (closest-hit + shadow rays traced by all ranks in the K timed steps) / wall time, in
Mrays/s, with barrier + torch.cuda.synchronize() on both sides and the MAX over ranks.

Besides `value` (a THROUGHPUT figure: three frames in flight, batched samples) the line carries
  latency_ms  one frame alone (raytrace_n(view, 4) + synchronize), nothing else on the GPU;
  drop_in     what the unchanged caller gets (crates/standalone, app.rs:297-318; SURVEY §8d span): ONE renderer,
              4 x raytrace() + read_radiance() (33 MB device -> host) per frame, no raytrace_n;
  roofline    k_trace: algorithmic bytes per launch / the UN-OVERLAPPED launch time (HIP events on the renderer's stream,
              nothing co-running: what rocprofv3's serialised kernel trace sees) / 8 TB/s; `overlapped` = the same over the
              timed region, where a launch shares the CUs with two other frames; `limits` = the PMC-derived ceilings;
  cpu_baseline  the oracle ("port") on the host cores, bounded sample.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")  # read by the HIP runtime at initialisation: one hardware queue per stream
os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")  # single-node control plane; the box's hostname may not resolve

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

import loupiote_amd as lp  # noqa: E402

if not os.path.exists(lp.LIB_PATH):  # the built library normally travels with the tree; a fresh checkout builds it (hipcc, ~1 min)
    if int(os.environ.get("LOCAL_RANK", "0")) == 0:
        from loupiote_amd import build as _build
        _build.build()
    else:  # one builder per node; the other ranks wait for the file
        _t0 = time.time()
        while not os.path.exists(lp.LIB_PATH) and time.time() - _t0 < 900:
            time.sleep(1.0)
        time.sleep(2.0)
from loupiote_amd import scenes, testing as T  # noqa: E402

WIDTH, HEIGHT, SPP, DEPTH = 1920, 1080, 4, 8
FRAMES_PER_STEP = 10
HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


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


def cpu_baseline(desc, view, threads):
    """The CPU oracle ("port") timed on the host cores on a bounded sample of the same workload:
    whole 1920x1080 frames of 1 spp each (the same seeds the GPU frame uses for its 1st, 2nd ... sample),
    as many as fit in ~12 s of wall time."""
    from oracle import orc  # checker only: the baseline leg, never the product path
    from oracle import harness
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
    return {"value": rays / secs / 1e6, "unit": "Mrays/s", "cores": threads, "kind": "port", "nproc": os.cpu_count(), "cpu": cpu, "cpu_quota": "threads = the cgroup CPU quota (cpu.max) when there is one",
            "per_thread": rays / secs / 1e6 / max(threads, 1),
            "sample": "oracle/lpt_oracle.c (scalar C, persistent pthread pool, SAH BVH2, 16x16 tiles): %d whole 1-spp frames of the same 1920x1080 depth-8 "
                      "workload with the seeds of the GPU frame's 1st, 2nd ... sample (%.1f Mrays in %.1f s); a reported baseline, not a target" % (frames, rays / 1e6, secs)}


def baseline_metric():
    """the metric string exactly as BASELINE.json spells it"""
    try:
        return json.load(open(os.path.join(ROOT, "BASELINE.json")))["metric"]
    except Exception:
        return "Mrays/s (+ ms/frame) at 1920\u00d71080, 4 spp, Sponza; 1/2/4/8 GPU"


def load_profile_json(name):
    path = os.path.join(ROOT, "profiles", name)
    if os.path.exists(path):
        try:
            return json.load(open(path))
        except Exception:
            return None
    return None


def main():
    global WIDTH, HEIGHT, SPP
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip latency / drop_in / solo measurements (experiments)")
    ap.add_argument("--scene", default="atrium")
    ap.add_argument("--width", type=int, default=WIDTH, help="experiments only; the reported config is the default")
    ap.add_argument("--height", type=int, default=HEIGHT)
    ap.add_argument("--spp", type=int, default=SPP)
    ap.add_argument("--frames-per-step", type=int, default=FRAMES_PER_STEP)
    ap.add_argument("--emulate-shard", type=int, default=0, help="experiments: render only rank 0's tiles of an N-way shard on one GPU")
    ap.add_argument("--force-dist", action="store_true", help="take the N>1 code path (process group, communicator, exchange) even with one rank")
    ap.add_argument("--exchange", choices=["gather", "reduce"], default="gather",
                    help="frame exchange for N>1: owned tiles only (W*H/N*16 B per rank, grouped send/recv) or the dense ncclReduce of the accumulation buffer")
    ap.add_argument("--pipeline", type=int, default=0, help="renderers (each with its own HIP stream) that take consecutive frames in turn; "
                    "default 3 on one GPU, 4 for tile shards (a 1/8 shard: 1.79 ms per frame with 3, 1.70 with 4, 1.71 with 6, 1.84 with 8 in flight)")
    ap.add_argument("--no-batch", action="store_true", help="4 separate raytrace() calls instead of raytrace_n(view, 4)")
    args = ap.parse_args()
    WIDTH, HEIGHT, SPP = args.width, args.height, args.spp
    BATCH = not args.no_batch
    FPS = max(1, args.frames_per_step)

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node %d" % args.gpus)
    torch.cuda.set_device(local_rank)
    use_dist = world > 1 or args.force_dist
    if use_dist:
        if "MASTER_ADDR" not in os.environ:
            os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", "29517"
        dist.init_process_group("gloo", rank=rank, world_size=world)  # control plane only (id rendezvous, barriers, max over ranks)

    dev = lp.Device(local_rank)
    comm = None
    if use_dist:
        box = [lp.Comm.unique_id() if rank == 0 else None]
        dist.broadcast_object_list(box, src=0)
        comm = lp.Comm(dev, box[0], rank, world)   # ncclCommInitRank inside the library (RCCL over xGMI)
    xmode = lp.EXCHANGE_REDUCE if args.exchange == "reduce" else lp.EXCHANGE_GATHER_TILES
    desc = scenes.synthetic_atrium(textures=not os.environ.get("LPT_BENCH_NOTEX"))
    scene = scenes.to_product(desc)
    sg = lp.SceneGPU.new_from_scene(scene, dev, gpu_build=bool(os.environ.get("LPT_BENCH_GPU_BUILD")))
    probe = lp.ProbeGPU(dev, desc["probe"], desc["probe"].shape[1], desc["probe"].shape[0])
    view = T.look(desc["camera"]["origin"], desc["camera"]["direction"])
    P = args.pipeline if args.pipeline > 0 else (4 if (world > 1 or args.emulate_shard > 1) else 3)

    def make_renderer(lanes=1):
        rr = lp.Renderer(dev, (WIDTH, HEIGHT))
        if lanes:
            rr.set_lanes(lanes)          # the throughput loop overlaps frames of DIFFERENT renderers: one wavefront lane each
        rr.downsample_factor = 1.0
        rr.resize(dev, sg, probe, (WIDTH, HEIGHT))
        rr.set_max_bounces(DEPTH)
        rr.set_vfov(T.VFOV)
        if comm is not None:
            rr.set_comm(comm)                      # = set_shard(rank, world, 32, 8) + the binding
            rr.set_resources(dev, sg, probe)
        elif args.emulate_shard > 1:
            rr.set_shard(0, args.emulate_shard, 32, 8)
            rr.set_resources(dev, sg, probe)
        return rr

    rs = [make_renderer() for _ in range(P)]
    frame_no = [0]

    def frame(r=None):
        if r is None:
            r = rs[frame_no[0] % P]
            frame_no[0] += 1
        r.reset_accumulation()
        r.accumulate = True
        if BATCH:
            r.raytrace_n(view, SPP)      # == SPP x { raytrace(view); accumulate = true } as one wavefront
        else:
            for _ in range(SPP):
                r.raytrace(view)
        if comm is not None:
            r.exchange(xmode)            # RCCL on the renderer's stream, behind the frame's kernels; rank 0 presents the frame

    def step():
        for _ in range(FPS):
            frame()

    def fence():
        if use_dist:
            dist.barrier()
        for rr in rs:
            rr.synchronize()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    for rr in rs:
        rr.reset_ray_counts()
        rr.enable_timings(True)
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    timings = {}
    closest_l = shadow_l = shaded_l = 0
    for rr in rs:
        for k, v in rr.timings().items():
            t0_, n0_ = timings.get(k, (0.0, 0))
            timings[k] = (t0_ + v[0], n0_ + v[1])
        rr.enable_timings(False)
        c_ = rr.ray_counts()
        closest_l += c_.closest; shadow_l += c_.shadow; shaded_l += c_.shaded

    tl = torch.tensor([elapsed], dtype=torch.float64)
    rays = torch.tensor([closest_l, shadow_l, shaded_l], dtype=torch.float64)
    if use_dist:
        dist.all_reduce(tl, op=dist.ReduceOp.MAX)
        dist.all_reduce(rays, op=dist.ReduceOp.SUM)
    elapsed = float(tl.item())
    closest, shadow, shaded = [float(x) for x in rays.tolist()]
    n_frames = args.steps * FPS

    # ---- roofline of the dominant kernel (k_trace), this rank: algorithmic bytes per launch
    # = closest rays * (32 B ray read + 16 B hit write + N*80 B nodes + T*48 B triangles)
    # + shadow rays * (32 B ray read + 4 B + Ns*80 B + Ts*48 B), with N, T, Ns, Ts (mean nodes visited /
    # triangles tested per ray) measured by the stats variant of the same kernel on the same frame,
    # outside the timed region.
    r = rs[0]
    r.enable_stats(True)
    r.reset_ray_counts()
    frame(r)
    fence()
    st = r.ray_counts()
    r.enable_stats(False)
    n_bar = st.nodes / max(st.closest, 1)
    t_bar = st.tris / max(st.closest, 1)
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

    o_avg, o_launches, o_bytes, o_achieved = trace_stage(timings, closest_l, shadow_l)
    rays_per_launch = (closest_l + shadow_l) / max(o_launches, 1)
    extras = not args.no_extras
    # the same launches with nothing co-running: SOLO_FRAMES frames on renderer 0 alone (HIP events on its stream)
    SOLO_FRAMES = 3
    r.reset_ray_counts()
    r.enable_timings(True)
    for _ in range(SOLO_FRAMES):
        frame(r)
        fence()
    solo_t = r.timings()
    r.enable_timings(False)
    sc_ = r.ray_counts()
    s_avg, s_launches, s_bytes, s_achieved = trace_stage(solo_t, sc_.closest, sc_.shadow)

    # N>1 (or --force-dist): rank 0's PRESENTED frame after an exchange must hold every pixel of the frame with its 4 samples
    exchange_ok = None
    if comm is not None:
        frame(r)
        fence()
        if rank == 0:
            img = r.read_radiance()
            exchange_ok = bool(np.all(img[..., 3] == 1.0) and np.all(np.isfinite(img)) and float(img[..., :3].mean()) > 0.0)

    latency = drop_in = None
    if extras:
        # ---- latency: one frame alone, host call to completion
        lat = []
        for _ in range(7):
            fence()
            t1 = time.perf_counter()
            frame(r)
            r.synchronize()
            lat.append((time.perf_counter() - t1) * 1e3)
        lat.sort()
        latency = {"min": lat[0], "median": lat[len(lat) // 2], "frames": len(lat),
                   "what": "one frame alone: reset_accumulation + raytrace_n(view, 4)%s + stream synchronize" % (" + exchange" if comm is not None else "")}
        # ---- drop-in: the unchanged caller's protocol on ONE renderer (SURVEY §8d span: raytrace() x spp ... read_radiance())
        if comm is None and args.emulate_shard <= 1:
            DROP_FRAMES = 10
            fence()
            rd = make_renderer(lanes=0)          # a renderer as the library hands it out (default: 2 wavefront lanes)
            for _ in range(2):                   # warm-up: the lanes allocate their ray buffers on first use
                for _ in range(SPP):
                    rd.raytrace(view)
            rd.synchronize()
            rd.reset_ray_counts()
            t1 = time.perf_counter()
            for _ in range(DROP_FRAMES):
                rd.reset_accumulation()
                rd.accumulate = True
                for _ in range(SPP):
                    rd.raytrace(view)
                img = rd.read_radiance()         # blocking; 33 MB device -> host inside the span
            dt = time.perf_counter() - t1
            dc = rd.ray_counts()
            rd.close()
            drop_in = {"ms_per_frame": dt / DROP_FRAMES * 1e3, "value": (dc.closest + dc.shadow) / dt / 1e6, "unit": "Mrays/s", "frames": DROP_FRAMES,
                       "what": "ONE renderer with the library's defaults, per frame: reset_accumulation; 4 x raytrace(view) (no raytrace_n); read_radiance() "
                               "(k_resolve + 33 MB D2H into pageable host memory) — the span SURVEY §8d defines and crates/standalone issues (app.rs:297-318).  "
                               "The renderer's two wavefront lanes let consecutive raytrace() calls overlap (18.2 ms with one lane)",
                       "checksum": float(np.float64(img[..., :3].sum()))}

    if rank == 0:
        traffic_j = load_profile_json("traffic.json") or {}
        limits_j = load_profile_json("limits.json")
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
            "data": "synthetic",
            "config": {"workload": "synthetic_atrium(seed=2) [Sponza stand-in, 262144 tris], 1920x1080, 4 spp, depth 8, "
                                   "camera (-10,1,0)->(1,0.35,0); frame = raytrace_n(view,4) == 4 x raytrace%s; step = %d frames, %d frames in flight "
                                   "(throughput; see latency_ms and drop_in for one frame alone / the unbatched protocol with read-back)"
                                   % (" + lpt_renderer_exchange(%s)" % args.exchange if use_dist else "", FPS, P),
                       "frames_per_step": FPS, "frames_timed": n_frames, "timed_region_s": elapsed,
                       "tiles": "32x8 interleaved, tile_id mod N", "exchange": (args.exchange + " (native RCCL, lpt_renderer_exchange)") if use_dist else "none", "exchange_frame_complete_on_rank0": exchange_ok,
                       "rays_per_frame": (closest + shadow) / n_frames, "rays_per_step": (closest + shadow) / args.steps,
                       "closest_rays": closest, "shadow_rays": shadow, "shaded_hits": shaded},
            "ms_per_frame": elapsed / n_frames * 1e3,
            "latency_ms": latency,
            "drop_in": drop_in,
            "roofline": {"bound": "hbm", "kernel": "k_trace", "achieved": s_achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": s_achieved / HBM_PEAK_GBS,
                         "traffic": traffic_j.get("k_trace_bytes_per_launch"), "traffic_source": traffic_j.get("source"),
                         "avg_launch_ms": s_avg, "launches": s_launches, "bytes_per_launch": s_bytes,
                         "basis": "un-overlapped launches: %d frames on one renderer with nothing co-running, HIP events on its stream around every k_trace launch — "
                                  "the duration rocprofv3's kernel trace reports for the same launches (profiles/, *_solo_kernel_stats.csv)" % SOLO_FRAMES,
                         "overlapped": {"achieved": o_achieved, "frac": o_achieved / HBM_PEAK_GBS, "avg_launch_ms": o_avg, "launches": o_launches, "bytes_per_launch": o_bytes,
                                        "frames_in_flight": P,
                                        "note": "the timed region: HIP events around every k_trace launch while %d frames share the chip; a launch then also counts the time it "
                                                "waits for, or shares, the CUs — a scheduling figure, not a kernel figure" % P},
                         "region": {"achieved": (closest_l * b_ray + shadow_l * b_sh) / elapsed / 1e9, "frac": (closest_l * b_ray + shadow_l * b_sh) / elapsed / 1e9 / HBM_PEAK_GBS,
                                    "note": "all k_trace algorithmic bytes of the timed region / its wall time (which also contains k_shade, ray generation, accumulation): a lower bound"},
                         "limits": limits_j,
                         "binding_limit": ({"name": "valu_issue", "frac": limits_j["valu_issue"]["frac"], "source": limits_j.get("source"),
                                            "note": "the kernel's algorithmic bytes come from L2 / Infinity Cache (traffic is 0.31 of the HBM roof): what binds it is the VALU issue "
                                                    "rate of its instruction mix (PMC pass, tools/limits_from_pmc.py), not the HBM roof `frac` is quoted against"}
                                           if limits_j and "valu_issue" in limits_j else None),
                         "rays_per_launch": rays_per_launch, "bytes_per_ray": b_ray, "bytes_per_shadow_ray": b_sh, "nodes_per_ray": n_bar, "tris_per_ray": t_bar,
                         "shadow_nodes_per_ray": ns_bar, "shadow_tris_per_ray": ts_bar,
                         "wave": {"live_lanes_per_step": st.live_lanes / max(st.wave_steps, 1), "node_lanes_per_step": st.node_lanes / max(st.wave_steps, 1),
                                  "tri_lanes_per_step": st.tri_lanes / max(st.wave_steps, 1), "lane_slots_per_ray": 64.0 * st.wave_steps / max(st.closest, 1)}},
            "stage_ms_per_frame": {k: v[0] / n_frames for k, v in timings.items()},
            "stage_ms_per_frame_solo": {k: v[0] / SOLO_FRAMES for k, v in solo_t.items()},
            "accel": {"triangles": accel.triangles, "nodes": accel.nodes, "node_bytes": accel.node_bytes,
                      "tri_bytes": accel.tri_bytes, "depth": accel.max_depth, "build_ms": accel.build_ms},
        }
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(desc, view, effective_cpus())
        elif world > 1:
            out["cpu_baseline"] = None
    else:
        out = None
    if use_dist:
        dist.barrier()
    for rr in rs:
        rr.close()
    if comm is not None:
        comm.close()
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


if __name__ == "__main__":
    main()
