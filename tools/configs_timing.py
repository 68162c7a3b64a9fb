#!/usr/bin/env python3
"""Secondary measurements: the other BASELINE.json configs on one MI355X (they are parity cases, not the bench line).
usage (GPU box): PYTHONPATH=. python tools/configs_timing.py   -> one JSON object per config"""
import json
import os
import sys
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401,E402  (before libloupiote_hip.so: see INTEGRATION.md §7)

import loupiote_amd as lp  # noqa: E402
from loupiote_amd import scenes, testing as T  # noqa: E402


ONLY = [a for a in sys.argv[1:] if not a.startswith("-")]   # substrings of the config names to run (default: all)


def measure(dev, name, scene, probe, cam, w, h, spp, depth, frames=12, warm=3, mode=lp.BlitMode.Pahtrace, pipeline=3, stats=False, shard=None):
    if ONLY and not any(o in name for o in ONLY):
        return
    if callable(scene):
        scene = scene()
    """stats: one more frame with the stats kernels — 8-wide nodes visited / triangles tested per ray of the per-ray traversal launches (what a builder's tree costs on
    this triangle distribution).  shard = (rank, world): that rank's tile shard of the frame alone (strong scaling of one frame, emulated on one GPU)."""
    sg = lp.SceneGPU.new_from_scene(scene, dev)
    pr = lp.ProbeGPU(dev, probe, probe.shape[1], probe.shape[0])
    view = T.look(*cam)
    rs = []
    for _ in range(pipeline):
        r = lp.Renderer(dev, (w, h))
        r.downsample_factor = 1.0
        r.resize(dev, sg, pr, (w, h))
        r.set_max_bounces(depth)
        r.set_vfov(T.VFOV)
        r.set_blit_mode(mode)
        if shard:
            r.set_shard(shard[0], shard[1], 32, 8)
            r.set_resources(dev, sg, pr)
        rs.append(r)

    def step(k):
        r = rs[k % pipeline]
        r.reset_accumulation()
        r.accumulate = True
        r.raytrace_n(view, spp)

    for k in range(warm):
        step(k)
    for r in rs:
        r.synchronize(); r.reset_ray_counts()
    dt, rays = None, 0
    for _ in range(3 if w * h < 500000 else 1):   # small frames: the best of three passes (a pass right behind the create / destroy of other renderers is sometimes 3-5 x slower)
        for r in rs:
            r.reset_ray_counts()
        t0 = time.perf_counter()
        for k in range(frames):
            step(k)
        for r in rs:
            r.synchronize()
        d1 = time.perf_counter() - t0
        if dt is None or d1 < dt:
            dt, rays = d1, sum(r.ray_counts().closest + r.ray_counts().shadow for r in rs)
    out = {"config": name, "size": [w, h], "spp": spp, "depth": depth, "ms_per_frame": dt / frames * 1e3, "Mrays_per_s": rays / dt / 1e6,
           "triangles": sg.stats().triangles, "nodes": sg.stats().nodes, "tree_depth": sg.stats().max_depth, "frames_in_flight": pipeline}
    if shard:
        out["shard"] = list(shard)
    if stats:
        r = rs[0]
        r.synchronize()
        r.enable_stats(True)
        r.reset_ray_counts()
        step(0)
        r.synchronize()
        c = r.ray_counts()
        r.enable_stats(False)
        per_ray = max(c.closest - c.primary, 1)
        out.update({"nodes_per_ray": c.nodes / per_ray, "tris_per_ray": c.tris / per_ray, "shadow_nodes_per_ray": c.shadow_nodes / max(c.shadow, 1),
                    "shadow_tris_per_ray": c.shadow_tris / max(c.shadow, 1), "rays_per_frame": c.closest + c.shadow})
    for r in rs:
        r.close()
    pr.close(); sg.close()
    print(json.dumps(out), flush=True)


def main():
    dev = lp.Device(0)
    glb = open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "cornell-box.glb"), "rb").read()
    s = lp.Scene(); lp.loaders.load_gltf(glb, s); s.set_light(0, T.cornell_light())
    measure(dev, "1: cornell-box 256x256 1spp depth 4 (BASELINE configs[0]: the reference's CPU-runnable case), frames back to back on one renderer, no read-back", s, T.CORNELL_PROBE, (T.CORNELL_EYE, T.CORNELL_DIR), 256, 256, 1, 4, frames=40, pipeline=1)
    measure(dev, "2: cornell-box 1024x1024 4spp depth 8", s, T.CORNELL_PROBE, (T.CORNELL_EYE, T.CORNELL_DIR), 1024, 1024, 4, 8)
    d = scenes.synthetic_helmet()
    measure(dev, "3: synthetic_helmet (DamagedHelmet stand-in) + sky probe 1920x1080 8spp depth 8", scenes.to_product(d), d["probe"],
            (d["camera"]["origin"], d["camera"]["direction"]), 1920, 1080, 8, 8)
    d = scenes.synthetic_atrium()
    sc = scenes.to_product(d)
    cam = (d["camera"]["origin"], d["camera"]["direction"])
    measure(dev, "4: synthetic_atrium 1920x1080 4spp depth 8 (the bench line)", sc, d["probe"], cam, 1920, 1080, 4, 8, stats=True)
    measure(dev, "tiny: synthetic_atrium 64x36 4spp depth 8, frames back to back on one renderer, no read-back (a wave per ray, LPT_OPT_COOP_RAYS)", sc, d["probe"], cam, 64, 36, 4, 8, frames=40, pipeline=1)
    measure(dev, "small: synthetic_atrium 384x216 4spp depth 8, frames back to back on one renderer, no read-back (per-bounce launches, tails in place)", sc, d["probe"], cam, 384, 216, 4, 8, frames=40, pipeline=1)
    measure(dev, "5a: synthetic_atrium 3840x2160 64spp progressive (8 x raytrace_n(8)) depth 8", sc, d["probe"], cam, 3840, 2160, 8, 8, frames=8, warm=2, pipeline=2)
    measure(dev, "5b: synthetic_atrium 3840x2160 temporal accumulate, 1 spp per frame, depth 8", sc, d["probe"], cam, 3840, 2160, 1, 8, frames=16, warm=3,
            mode=lp.BlitMode.Temporal, pipeline=1)
    # config 5 across GPUs, emulated — rank 0's 1/8 tile shard of the 64-spp 4K frame alone (fused wavefronts of up to 64 samples): against 5a's whole
    # frame this is the strong scaling 8 GPUs can give before any exchange
    measure(dev, "5a whole, one renderer: synthetic_atrium 3840x2160 64spp (8 x raytrace_n(8)) depth 8", sc, d["probe"], cam, 3840, 2160, 8, 8, frames=8, warm=2, pipeline=1)
    measure(dev, "5a shard 1/8 (rank 0 of 8), one renderer: the same frame's tile shard alone", sc, d["probe"], cam, 3840, 2160, 8, 8, frames=8, warm=2, pipeline=1, shard=(0, 8))
    # Sponza-shaped triangle distributions (the stand-in's triangles are uniformly small): nodes / triangles per ray and the frame time
    dq = scenes.synthetic_atrium(shell_quads=True)
    measure(dev, "hall_large: synthetic_atrium(shell_quads=True) — floor / walls / roof as two triangles each — 1920x1080 4spp depth 8", scenes.to_product(dq), dq["probe"],
            (dq["camera"]["origin"], dq["camera"]["direction"]), 1920, 1080, 4, 8, stats=True)
    dh = scenes.synthetic_hall()
    measure(dev, "hall: synthetic_hall (two-triangle walls around centimetre gravel, slivers, coincident quads, a telescope) 1920x1080 4spp depth 8", scenes.to_product(dh), dh["probe"],
            (dh["camera"]["origin"], dh["camera"]["direction"]), 1920, 1080, 4, 8, stats=True)


if __name__ == "__main__":
    sys.exit(main())
