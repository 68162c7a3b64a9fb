#!/usr/bin/env python3
"""Generates tests/golden/cornell_256_d4_s1.npz from the CPU oracle (BASELINE config 1:
cornell-box.glb, 256x256, 1 spp, depth 4, build-defined camera/light/probe of loupiote_amd.testing).

The reference cannot produce vectors for this path (its integrator is the absent albedo_rtx crate and
it has no headless mode — SURVEY.md §8c), so the committed vectors pin the ORACLE, which in turn is
pinned by the analytic known-answer tests in tests/test_oracle_kat.py.  Run from the repo root."""
import hashlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from loupiote_amd import testing as T  # noqa: E402,F401
from oracle import harness  # noqa: E402

glb = open(os.path.join(ROOT, "tests", "golden", "cornell-box.glb"), "rb").read()
img, cnt = harness.render_oracle(glb, 256, 256, 4, 1)
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "cornell_256_d4_s1.npz"),
                    crop=img[96:160, 96:160], sha256=hashlib.sha256(img.tobytes()).hexdigest(),
                    mean=img[..., :3].mean(axis=(0, 1)), counts=np.array([cnt.closest, cnt.shadow, cnt.shaded], np.int64))
print("closest %d shadow %d shaded %d sha256 %s" % (cnt.closest, cnt.shadow, cnt.shaded, hashlib.sha256(img.tobytes()).hexdigest()))
