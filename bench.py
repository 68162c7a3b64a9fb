#!/usr/bin/env python3
"""bench.py — Mrays/s of the path-tracing hot path on N MI355X (one process per GPU).

Workload (BASELINE.json metric: "Mrays/s (+ ms/frame) at 1920x1080, 4 spp, Sponza"):
  synthetic_atrium(seed=2) — the Sponza STAND-IN (the real asset is absent) —
  1920x1080, 4 spp (= 4 raytrace() calls with accumulate), path depth 8, camera = the reference's
  start pose.  One "step" = one such frame: reset_accumulation(); accumulate=true;
  4 x Renderer::raytrace(view) — issued as lpt_renderer_raytrace_n(view, 4), the bit-identical batched form; for N>1 an RCCL reduce(sum) of the radiance buffer to rank 0.
  Inputs (scene, BVH, probe, textures) are resident in HBM before the timed region.
  Consecutive steps rotate over three renderers with their own HIP streams (--pipeline 3: three frames in flight,
  triple buffering), so the tail of frame k (and its collective) overlaps the heads of the next frames; every step is
  still one complete frame.  GPU_MAX_HW_QUEUES is raised to 8 (ROCm default 4, of which the streams here got two):
  with fewer hardware queues than streams the frames serialise again.
  N>1: frames shard by interleaved 32x8 pixel tiles (tile id mod N), per-GPU work shrinks as
  N grows ("strong" scaling of one frame).

value = (closest-hit + shadow rays traced by all ranks in the K timed steps) / wall time, in
Mrays/s, with barrier + torch.cuda.synchronize() on both sides and the MAX over ranks.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")  # read by the HIP runtime at initialisation: one hardware queue per stream

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
HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


class _DevBuf:
    """exposes a raw device pointer to torch (plumbing for torch.distributed only)"""

    def __init__(self, ptr, nbytes):
        self.__cuda_array_interface__ = {"shape": (nbytes // 4,), "typestr": "<f4", "data": (ptr, False), "version": 2}


def cpu_baseline(desc, view, threads):
    """The CPU oracle ("port") timed on the host cores on a bounded sample of the same workload:
    whole 1920x1080 frames of 1 spp each (the same seeds the GPU step uses for its 1st, 2nd ... sample),
    as many of the 4 as fit in ~15 s of wall time."""
    from oracle import orc  # checker only: the baseline leg, never the product path
    from oracle import harness
    s = harness.to_oracle(desc)
    sc = orc.OracleScene.from_scene(s, probe=desc["probe"])
    rays, secs, frames = 0, 0.0, 0
    while frames < SPP and secs < 15.0:
        t0 = time.perf_counter()
        _, cnt = sc.render(WIDTH, HEIGHT, view, T.VFOV, DEPTH, frames=1, seed_counter=frames * DEPTH, threads=threads, want_counters=True)
        secs += time.perf_counter() - t0
        rays += cnt.closest + cnt.shadow
        frames += 1
    return {"value": rays / secs / 1e6, "unit": "Mrays/s", "cores": threads, "kind": "port",
            "sample": "oracle/lpt_oracle.c (scalar C, pthreads, 16x16 tiles), %d of the %d spp of the same 1920x1080 depth-8 frame "
                      "(%.1f Mrays in %.1f s)" % (frames, SPP, rays / 1e6, secs)}


def main():
    global WIDTH, HEIGHT, SPP
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--scene", default="atrium")
    ap.add_argument("--width", type=int, default=WIDTH, help="experiments only; the reported config is the default")
    ap.add_argument("--height", type=int, default=HEIGHT)
    ap.add_argument("--spp", type=int, default=SPP)
    ap.add_argument("--emulate-shard", type=int, default=0, help="experiments: render only rank 0's tiles of an N-way shard on one GPU")
    ap.add_argument("--force-dist", action="store_true", help="take the N>1 code path (process group, reduce) even with one rank")
    ap.add_argument("--exchange", choices=["reduce", "gather"], default="reduce",
                    help="frame exchange for N>1: dense RCCL reduce of the accumulation buffer (north star) or a gather of owned tiles only")
    ap.add_argument("--pipeline", type=int, default=3, help="renderers (each with its own HIP stream) that take consecutive steps in turn")
    ap.add_argument("--no-batch", action="store_true", help="4 separate raytrace() calls instead of raytrace_n(view, 4)")
    args = ap.parse_args()
    WIDTH, HEIGHT, SPP = args.width, args.height, args.spp
    BATCH = not args.no_batch

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
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    dev = lp.Device(local_rank)
    desc = scenes.synthetic_atrium(textures=not os.environ.get("LPT_BENCH_NOTEX"))
    scene = scenes.to_product(desc)
    sg = lp.SceneGPU.new_from_scene(scene, dev, gpu_build=bool(os.environ.get("LPT_BENCH_GPU_BUILD")))
    probe = lp.ProbeGPU(dev, desc["probe"], desc["probe"].shape[1], desc["probe"].shape[0])
    view = T.look(desc["camera"]["origin"], desc["camera"]["direction"])
    P = max(1, args.pipeline)
    rs, exts, accums, frames = [], [], [], []
    for _ in range(P):
        rr = lp.Renderer(dev, (WIDTH, HEIGHT))
        rr.downsample_factor = 1.0
        rr.resize(dev, sg, probe, (WIDTH, HEIGHT))
        rr.set_max_bounces(DEPTH)
        rr.set_vfov(T.VFOV)
        if world > 1 or args.emulate_shard > 1:
            rr.set_shard(rank, max(world, args.emulate_shard), 32, 8)
            rr.set_resources(dev, sg, probe)
        rs.append(rr)
        exts.append(torch.cuda.ExternalStream(rr.stream(), device=torch.device("cuda", local_rank)))
        ptr, nbytes = rr.radiance_device_ptr()
        accums.append(torch.as_tensor(_DevBuf(ptr, nbytes), device=torch.device("cuda", local_rank)))
        frames.append(torch.empty_like(accums[-1]) if use_dist else None)
    gathers = None
    if use_dist and args.exchange == "gather":
        from loupiote_amd.dist import OwnedTileGather
        gathers = [OwnedTileGather(WIDTH, HEIGHT, rank, world, device=torch.device("cuda", local_rank)) for _ in range(P)]
    r = rs[0]
    step_no = [0]

    def step():
        k = step_no[0] % P
        step_no[0] += 1
        r = rs[k]
        r.reset_accumulation()
        r.accumulate = True
        if BATCH:
            r.raytrace_n(view, SPP)      # == SPP x { raytrace(view); accumulate = true } as one wavefront
        else:
            for _ in range(SPP):
                r.raytrace(view)
        if use_dist:
            # radiance reduce over xGMI: ordered after this renderer's stream, which its next
            # frame's kernels in turn wait on (torch issues the RCCL op relative to the stream)
            with torch.cuda.stream(exts[k]):
                if gathers is not None:
                    gathers[k](accums[k].view(HEIGHT, WIDTH, 4))   # owned pixels only (W*H/N * 16 B per rank)
                else:
                    frames[k].copy_(accums[k], non_blocking=True)
                    dist.reduce(frames[k], dst=0, op=dist.ReduceOp.SUM)

    def fence():
        if use_dist:
            dist.barrier()
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

    class _C:
        closest, shadow, shaded = closest_l, shadow_l, shaded_l
    counts = _C

    tl = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
    rays = torch.tensor([counts.closest, counts.shadow, counts.shaded], dtype=torch.float64, device="cuda")
    if use_dist:
        dist.all_reduce(tl, op=dist.ReduceOp.MAX)
        dist.all_reduce(rays, op=dist.ReduceOp.SUM)
    elapsed = float(tl.item())
    closest, shadow, shaded = [float(x) for x in rays.tolist()]

    # ---- roofline of the dominant kernel (k_trace), rank 0: algorithmic bytes per launch
    # = closest rays * (32 B ray read + 16 B hit write + N*80 B nodes + T*48 B triangles)
    # + shadow rays * (32 B ray read + 4 B + Ns*80 B + Ts*48 B), with N, T, Ns, Ts (mean nodes visited /
    # triangles tested per ray) measured by the stats variant of the same kernel on the same frames,
    # outside the timed region.
    r = rs[0]
    step_no[0] = 0
    r.enable_stats(True)
    r.reset_ray_counts()
    step()
    fence()
    st = r.ray_counts()
    r.enable_stats(False)
    n_bar = st.nodes / max(st.closest, 1)
    t_bar = st.tris / max(st.closest, 1)
    ns_bar = st.shadow_nodes / max(st.shadow, 1)
    ts_bar = st.shadow_tris / max(st.shadow, 1)
    accel = sg.stats()
    # dominant kernel: k_trace (closest-hit rays of bounce b+1 and shadow rays of bounce b in one persistent launch;
    # stages "intersection" = launches that carry closest-hit rays, "shadow" = the last, shadow-only launch)
    b_ray = 32.0 + 16.0 + n_bar * accel.node_bytes + t_bar * accel.tri_bytes
    b_sh = 32.0 + 4.0 + ns_bar * accel.node_bytes + ts_bar * accel.tri_bytes

    def trace_stage(tm, cl, sh):
        ms = tm.get("intersection", (0.0, 0))[0] + tm.get("shadow", (0.0, 0))[0]
        launches = tm.get("intersection", (0.0, 0))[1] + tm.get("shadow", (0.0, 0))[1]
        avg = ms / max(launches, 1)
        byts = (cl * b_ray + sh * b_sh) / max(launches, 1)
        return avg, launches, byts, (byts / (avg * 1e-3) / 1e9 if avg > 0 else 0.0)

    avg_ms, i_launches, bytes_per_launch, achieved = trace_stage(timings, counts.closest, counts.shadow)
    rays_per_launch = (counts.closest + counts.shadow) / max(i_launches, 1)
    # the same kernel with nothing co-running (one extra untimed step on renderer 0): with --pipeline 2 the
    # kernels of two frames share the chip, which lengthens each launch although the step gets shorter
    r.reset_ray_counts()
    r.enable_timings(True)
    step_no[0] = 0
    step()
    fence()
    solo_t = r.timings()
    r.enable_timings(False)
    sc_ = r.ray_counts()
    s_avg, s_launches, _, solo = trace_stage(solo_t, sc_.closest, sc_.shadow)
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tpath):
        try:
            traffic = json.load(open(tpath)).get("k_trace_bytes_per_launch")
        except Exception:
            traffic = None

    if rank == 0:
        out = {
            "metric": "Mrays/s (+ ms/frame) at 1920x1080, 4 spp, Sponza; 1/2/4/8 GPU",
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
                                   "camera (-10,1,0)->(1,0.35,0); step = 1 frame (raytrace_n(view,4) == 4 x raytrace, + reduce)",
                       "tiles": "32x8 interleaved, tile_id mod N", "exchange": args.exchange if use_dist else "none", "rays_per_step": (closest + shadow) / args.steps,
                       "closest_rays": closest, "shadow_rays": shadow, "shaded_hits": shaded},
            "ms_per_frame": elapsed / args.steps * 1e3,
            "roofline": {"bound": "hbm", "kernel": "k_trace", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "avg_launch_ms": avg_ms, "launches": i_launches, "frames_in_flight": P,
                         "solo": {"achieved": solo, "frac": solo / HBM_PEAK_GBS, "avg_launch_ms": s_avg, "launches": s_launches}, "rays_per_launch": rays_per_launch,
                         "note": "avg_launch_ms / achieved / frac: HIP events around every k_trace launch of the timed region; with several frames in flight they include the time a launch shares or waits for the CUs (rocprofv3 serialises kernels, so its per-kernel average matches `solo`, the same kernel with nothing co-running)",
                         "bytes_per_launch": bytes_per_launch, "bytes_per_ray": b_ray, "bytes_per_shadow_ray": b_sh, "nodes_per_ray": n_bar, "tris_per_ray": t_bar,
                         "shadow_nodes_per_ray": ns_bar, "shadow_tris_per_ray": ts_bar,
                         "wave": {"live_lanes_per_step": st.live_lanes / max(st.wave_steps, 1), "node_lanes_per_step": st.node_lanes / max(st.wave_steps, 1),
                                  "tri_lanes_per_step": st.tri_lanes / max(st.wave_steps, 1), "lane_slots_per_ray": 64.0 * st.wave_steps / max(st.closest, 1)}},
            "stage_ms_per_step": {k: v[0] / args.steps for k, v in timings.items()},
            "accel": {"triangles": accel.triangles, "nodes": accel.nodes, "node_bytes": accel.node_bytes,
                      "tri_bytes": accel.tri_bytes, "depth": accel.max_depth, "build_ms": accel.build_ms},
        }
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(desc, view, os.cpu_count() or 1)
        elif world > 1:
            out["cpu_baseline"] = None
    else:
        out = None
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    for rr in rs:
        rr.close()
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
