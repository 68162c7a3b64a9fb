#!/usr/bin/env python3
"""Generates tests/golden/full_configs.npz: ORACLE vectors at the FULL sizes of BASELINE.json configs 3, 4 and 5
(SURVEY §8c "Golden vectors / fixtures"; config 2 as well), so that the -m gpu tests pin the HIP path to committed numbers at the
sizes the bench runs, not only to a live oracle at reduced sizes.

  cfg2  cornell-box.glb, 1024x1024, 4 spp, depth 8
  cfg3  DamagedHelmet stand-in + sky probe, 1920x1080, 8 spp, depth 8
  cfg4  Sponza stand-in (the bench workload), 1920x1080, 4 spp, depth 8
  cfg5p the same scene, 3840x2160, depth 8: the first 2 of the 64 progressive samples
  cfg5p64 the same, all 64 progressive samples (sha256, ray counts, means and windows of the 64-spp frame)
  cfg5t the same, BlitMode::Temporal, 2 frames (1 spp each): composite output, history
  cfg5d the same, BlitMode::DenoisedPathrace (temporal + 4 a-trous + composite), 2 frames: output, history

Per config: sha256 of the whole resolved float32 frame, exact ray counts (closest, shadow, shaded), per-channel
means, the oracle's own nodes / triangle tests per ray (its private BVH2; informational), and four 32x32 windows.
The reference cannot produce vectors for this path: these pin the ORACLE (itself pinned by
tests/test_oracle_kat.py), generated in the build container by this script.  Run from the repo root (~25 min on 8 cores, most of it the 64-spp 4K frame;
`--skip-64spp` keeps the committed cfg5p64 entries)."""
import hashlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from loupiote_amd import scenes, testing as T  # noqa: E402
from oracle import harness, orc  # noqa: E402

WINDOWS = lambda w, h: [(w // 2 - 16, h // 2 - 16), (w // 5, h // 4), (3 * w // 4, 2 * h // 3), (w // 3, 4 * h // 5)]  # noqa: E731


def record(out, key, img, cnt=None, extra=None):
    h, w = img.shape[:2]
    out[key + "_sha256"] = hashlib.sha256(np.ascontiguousarray(img).tobytes()).hexdigest()
    out[key + "_mean"] = img[..., :3].astype(np.float64).mean(axis=(0, 1))
    out[key + "_windows"] = np.array(WINDOWS(w, h), np.int32)
    out[key + "_crops"] = np.stack([img[y:y + 32, x:x + 32] for x, y in WINDOWS(w, h)])
    if cnt is not None:
        out[key + "_counts"] = np.array([cnt.closest, cnt.shadow, cnt.shaded], np.int64)
        out[key + "_oracle_nodes_tris_per_ray"] = np.array([cnt.nodes / max(cnt.closest + cnt.shadow, 1), cnt.tris / max(cnt.closest + cnt.shadow, 1)])
    for k, v in (extra or {}).items():
        out[key + "_" + k] = v
    print(key, out[key + "_sha256"][:16], out.get(key + "_counts"), flush=True)


def main():
    out = {}
    t0 = time.time()
    path = os.path.join(ROOT, "tests", "golden", "full_configs.npz")
    if "--only-cfg2" in sys.argv:   # add / refresh config 2 without re-rendering the big frames
        out = dict(np.load(path))
    # config 2: cornell-box.glb, 1024x1024, 4 spp, depth 8 (build-defined camera / light / probe of loupiote_amd.testing)
    glb = open(os.path.join(ROOT, "tests", "golden", "cornell-box.glb"), "rb").read()
    img, cnt = harness.render_oracle(glb, 1024, 1024, 8, 4)
    record(out, "cfg2", img, cnt)
    if "--only-cfg2" in sys.argv:
        np.savez_compressed(path, **out)
        print("done in %.0f s" % (time.time() - t0))
        return
    helmet = scenes.synthetic_helmet()
    osc = orc.OracleScene.from_scene(harness.to_oracle(helmet), probe=helmet["probe"])
    view = T.look(helmet["camera"]["origin"], helmet["camera"]["direction"])
    acc, cnt = osc.render(1920, 1080, view, T.VFOV, 8, frames=8, want_counters=True)
    record(out, "cfg3", orc.resolve(acc), cnt)
    del osc
    atrium = scenes.synthetic_atrium()
    osc = orc.OracleScene.from_scene(harness.to_oracle(atrium), probe=atrium["probe"])
    view = T.look(atrium["camera"]["origin"], atrium["camera"]["direction"])
    acc, cnt = osc.render(1920, 1080, view, T.VFOV, 8, frames=4, want_counters=True)
    record(out, "cfg4", orc.resolve(acc), cnt)
    acc, cnt = osc.render(3840, 2160, view, T.VFOV, 8, frames=2, want_counters=True)
    record(out, "cfg5p", orc.resolve(acc), cnt)
    den = orc.Denoiser(osc, 3840, 2160, T.VFOV, 8)
    for _ in range(2):
        img = den.frame(view, mode=2)
    _, _, _, hist = den.read()
    record(out, "cfg5t", img, None, {"history_sha256": hashlib.sha256(np.ascontiguousarray(hist).tobytes()).hexdigest()})
    del den
    den = orc.Denoiser(osc, 3840, 2160, T.VFOV, 8)
    for _ in range(2):
        img = den.frame(view, mode=1)   # BlitMode::DenoisedPathrace: temporal + a-trous x 4 + composite (asvgf.rs:278-290)
    _, _, rad, hist = den.read()
    record(out, "cfg5d", img, None, {"history_sha256": hashlib.sha256(np.ascontiguousarray(hist).tobytes()).hexdigest(),
                                     "radiance_sha256": hashlib.sha256(np.ascontiguousarray(rad).tobytes()).hexdigest()})
    del den
    if "--skip-64spp" in sys.argv and os.path.exists(path):
        old = dict(np.load(path))
        out.update({k: v for k, v in old.items() if k.startswith("cfg5p64_")})
    else:
        acc, cnt = osc.render(3840, 2160, view, T.VFOV, 8, frames=64, want_counters=True)
        record(out, "cfg5p64", orc.resolve(acc), cnt)
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "full_configs.npz"), **out)
    print("done in %.0f s" % (time.time() - t0))


if __name__ == "__main__":
    main()
