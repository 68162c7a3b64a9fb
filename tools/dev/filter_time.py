#!/usr/bin/env python3
"""On the GPU box: how long the ASVGF filter passes (temporal, 4 x a-trous, composite: launch_filter) take at 3840x2160 on ONE GPU,
beside the time one rank of an 8-way tile shard spends tracing the same frame (1 spp, depth 8) — is rank 0's whole-frame filter
what limits config 5 on 8 GPUs?  Prints one JSON line (kept under profiles/)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import loupiote_amd as lp  # noqa: E402
from loupiote_amd import scenes, testing as T  # noqa: E402

W, H, DEPTH, FRAMES = 3840, 2160, 8, 12


def main():
    dev = lp.Device(0)
    desc = scenes.synthetic_atrium()
    sg = lp.SceneGPU.new_from_scene(scenes.to_product(desc), dev)
    pr = lp.ProbeGPU(dev, desc["probe"], desc["probe"].shape[1], desc["probe"].shape[0])
    view = T.look(desc["camera"]["origin"], desc["camera"]["direction"])
    out = {"size": [W, H], "depth": DEPTH, "frames": FRAMES, "scene": desc["name"]}

    def run(mode, shard=None):
        r = lp.Renderer(dev, (W, H))
        r.downsample_factor = 1.0
        r.resize(dev, sg, pr, (W, H))
        r.set_max_bounces(DEPTH)
        r.set_vfov(T.VFOV)
        if shard:
            r.set_shard(shard[0], shard[1], 32, 8)
            r.set_resources(dev, sg, pr)
        r.set_blit_mode(mode)
        r.reset_accumulation()
        for _ in range(3):
            r.raytrace(view)
        r.synchronize()
        r.enable_timings(True)
        for _ in range(FRAMES):
            r.raytrace(view)
            r.synchronize()
        t = r.timings()
        r.close()
        return {k: v[0] / FRAMES for k, v in t.items() if v[1]}

    out["pathtrace_whole_frame_ms"] = run(lp.BlitMode.Pahtrace)
    out["temporal_whole_frame_ms"] = run(lp.BlitMode.Temporal)
    out["denoised_whole_frame_ms"] = run(lp.BlitMode.DenoisedPathrace)
    out["pathtrace_one_eighth_shard_ms"] = run(lp.BlitMode.Pahtrace, (0, 8))
    tr = lambda d: d.get("intersection", 0) + d.get("shadow", 0) + d.get("shading", 0) + d.get("ray generation", 0)  # noqa: E731
    out["summary"] = {"filter_temporal_ms": out["temporal_whole_frame_ms"].get("asvgf"), "filter_denoised_ms": out["denoised_whole_frame_ms"].get("asvgf"),
                      "trace_whole_frame_ms": tr(out["pathtrace_whole_frame_ms"]), "trace_one_eighth_shard_ms": tr(out["pathtrace_one_eighth_shard_ms"]),
                      "what": "asvgf = k_den_scatter + temporal (+ copy + 4 a-trous) + composite, HIP events on the renderer's stream; trace = ray generation + "
                              "k_trace + k_shade of one 1-spp depth-8 frame"}
    print(json.dumps(out))
    pr.close(); sg.close(); dev.close()


if __name__ == "__main__":
    main()
