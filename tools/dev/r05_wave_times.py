#!/usr/bin/env python3
"""When do the waves of a shard-sized traversal launch start, run dry, enter the tail and end?  Needs the LPT_EXP_WAVETIMES build:
  make -C loupiote_amd/csrc variant NAME=wt DEFS=-DLPT_EXP_WAVETIMES=1
  LPT_LIB_PATH=$PWD/loupiote_amd/libloupiote_hip_wt.so python tools/dev/r05_wave_times.py [tail_lanes]   (GPU box)"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: F401,E402

import loupiote_amd as lp  # noqa: E402
from loupiote_amd import _abi as A, scenes, testing as T  # noqa: E402

tail = int(sys.argv[1]) if len(sys.argv) > 1 else 4
waves_per_cu = int(sys.argv[2]) if len(sys.argv) > 2 else 0
dev = lp.Device(0)
desc = scenes.synthetic_atrium()
sg = lp.SceneGPU.new_from_scene(scenes.to_product(desc), dev)
pr = lp.ProbeGPU(dev, desc["probe"], desc["probe"].shape[1], desc["probe"].shape[0])
view = T.look(desc["camera"]["origin"], desc["camera"]["direction"])
r = lp.Renderer(dev, (1920, 1080))
r.downsample_factor = 1.0
r.resize(dev, sg, pr, (1920, 1080))
r.set_max_bounces(8)
r.set_vfov(T.VFOV)
r.set_option("tail_lanes", tail)
r.set_option("trace_waves_per_cu", waves_per_cu)
r.set_option("step_budget", 0)      # the probe's records live in the straggler list
r.set_shard(0, 8)
r.set_resources(dev, sg, pr)
for _ in range(4):
    r.reset_accumulation(); r.accumulate = True
    r.raytrace_n(view, 4)
    r.synchronize()
words = 9 * 8192 * 8
buf = np.zeros(words, np.uint32)
fn = A.lib().lpt_debug_read_wave_times
fn.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32]
fn.restype = C.c_int
assert fn(r._h, 0, buf.ctypes.data, words) == 0, lp._abi.last_error() if hasattr(lp._abi, "last_error") else "failed"
rec = buf.reshape(9, 8192, 8)
print("trace_waves_per_cu %d (0 = default)" % waves_per_cu)
print("tail_lanes %d; times in us from the launch's first wave start; one line per traversal launch (1..8)" % tail)
for L in range(1, 9):
    w = rec[L]
    w = w[w[:, 7] == 0x57415645]
    if not len(w):
        continue
    t0 = int(w[:, 0].min())
    us = lambda x: ((x.astype(np.int64) - t0) & 0xFFFFFFFF) / 100.0
    start, dry, end = us(w[:, 0]), us(w[:, 1]), us(w[:, 3])
    has_tail = w[:, 2] != 0
    tl = us(w[has_tail, 2]) if has_tail.any() else np.zeros(1)
    q = lambda a: "%.0f/%.0f/%.0f/%.0f" % (np.percentile(a, 5), np.percentile(a, 50), np.percentile(a, 95), a.max())
    print("launch %d: %d waves | start p5/p50/p95/max %s | dry %s | tail entry (%d waves) %s | end %s | lanes live at dry: mean %.1f | drain (end - dry) %s | tail phase (end - entry) %s"
          % (L, len(w), q(start), q(dry), int(has_tail.sum()), q(tl), q(end), w[:, 4].mean(), q(end - dry), q((us(w[has_tail, 3]) - tl) if has_tail.any() else np.zeros(1))))
    xcd = np.arange(8192)[rec[L][:, 7] == 0x57415645] & 7
    print("          per XCD (block & 7): dry median " + " ".join("%.0f" % np.median(dry[xcd == k]) for k in range(8)) + " | dry max " + " ".join("%.0f" % dry[xcd == k].max() for k in range(8))
          + " | end max " + " ".join("%.0f" % end[xcd == k].max() for k in range(8)))
    st = w[:, 6].astype(np.float64)
    ok = st > 0
    per = (end - dry)[ok] / st[ok]
    late = end[ok] > np.percentile(end[ok], 90)
    print("          lane-mode steps after dry: median %.0f max %.0f | us per step after dry (incl. the tail phase): median %.2f, of the last 10 %% of the waves to end %.2f | queue empty (first dry) at %.0f" % (np.median(st[ok]), st.max(), np.median(per), np.median(per[late]), dry.min()))
    # how much wave-time lies behind the moment the queues ran dry
    tot = float((end - start).sum())
    print("          wave-time: total %.0f wave-us, after dry %.0f (%.0f %%); launch ends at %.0f us, the median wave is dry at %.0f" % (tot, float((end - dry).sum()), 100.0 * float((end - dry).sum()) / tot, end.max(), np.median(dry)))
r.close(); pr.close(); sg.close(); dev.close()
