import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from loupiote_amd import scenes, testing as T
from oracle import orc, harness
print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
try:
    print("cpu.max", open("/sys/fs/cgroup/cpu.max").read().strip())
except Exception as e:
    print("cpu.max n/a", e)
desc = scenes.synthetic_atrium()
sc = orc.OracleScene.from_scene(harness.to_oracle(desc), probe=desc["probe"])
view = T.look(desc["camera"]["origin"], desc["camera"]["direction"])
for th in (8, 32, 64, 128, 256):
    t0 = time.time()
    acc, cnt = sc.render(1920, 1080, view, T.VFOV, 8, frames=1, threads=th, want_counters=True)
    dt = time.time() - t0
    r = cnt.closest + cnt.shadow
    print("threads %3d: %.2f s  %.2f Mrays/s  %.3f per thread" % (th, dt, r / dt / 1e6, r / dt / 1e6 / th), flush=True)
