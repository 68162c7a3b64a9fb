"""round 5: the pool kernel against the per-bounce launches, bit for bit, at rising sizes (first GPU runs of k_pool: small frames first).
usage: python tools/dev/r05_pool_check.py [max_stage]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import loupiote_amd as lp
from loupiote_amd import scenes, testing as T

def render(dev, sg, pr, desc, w, h, spp, opts, shard=None, stats=False):
    r = lp.Renderer(dev, (w, h))
    r.downsample_factor = 1.0
    r.resize(dev, sg, pr, (w, h))
    r.set_max_bounces(8)
    r.set_vfov(T.VFOV)
    for k, v in opts.items():
        r.set_option(k, v)
    if shard:
        r.set_shard(0, shard)
        r.set_resources(dev, sg, pr)
    if stats:
        r.enable_stats(True)
    view = T.look(desc["camera"]["origin"], desc["camera"]["direction"])
    out = None
    times = []
    for rep in range(3):
        r.reset_accumulation(); r.accumulate = True; r.reset_ray_counts()
        t0 = time.perf_counter()
        r.raytrace_n(view, spp)
        img = r.read_radiance()
        times.append((time.perf_counter() - t0) * 1e3)
        out = img
    c = r.ray_counts()
    q = r.queue_counts(8) if hasattr(r, "queue_counts") else None
    r.close()
    return out, c, min(times), q

def main():
    max_stage = int(sys.argv[1]) if len(sys.argv) > 1 else 9
    dev = lp.Device(0)
    desc = scenes.synthetic_atrium()
    scene = scenes.to_product(desc)
    sg = lp.SceneGPU.new_from_scene(scene, dev)
    pr = lp.ProbeGPU(dev, desc["probe"], desc["probe"].shape[1], desc["probe"].shape[0])
    PB = {"path_rays": 0, "pool_rays": 0}
    stages = [(64, 64, 1, None, {}), (64, 64, 4, None, {"packet_primary": 1}), (256, 256, 4, None, {"packet_primary": 1}), (256, 256, 1, None, {"packet_primary": 0}),
              (960, 540, 4, None, {}), (1920, 1080, 4, 8, {}), (1920, 1080, 4, 8, {"pool_waves": 8}), (1920, 1080, 4, 8, {"pool_waves": 4}), (1920, 1080, 4, None, {})]
    ok = True
    for i, (w, h, spp, shard, extra) in enumerate(stages[:max_stage]):
        ref, rc, rt, rq = render(dev, sg, pr, desc, w, h, spp, dict(PB, **{k: v for k, v in extra.items() if k.startswith("packet")}), shard)
        try:
            img, c, t, q = render(dev, sg, pr, desc, w, h, spp, dict({"path_rays": 0, "pool_rays": 0x7FFFFFFF}, **extra), shard)
        except Exception as e:
            print("stage %d %dx%d spp %d shard %s %s: FAILED %r" % (i, w, h, spp, shard, extra, e), flush=True)
            ok = False
            break
        same = np.array_equal(ref, img)
        cs = (rc.closest, rc.shadow, rc.shaded) == (c.closest, c.shadow, c.shaded)
        print("stage %d %dx%d spp %d shard %s %s: identical %s  counts %s (%d %d %d)  per-bounce %.3f ms  pool %.3f ms  maxdiff %g" %
              (i, w, h, spp, shard, extra, same, cs, c.closest, c.shadow, c.shaded, rt, t, float(np.abs(ref - img).max())), flush=True)
        if not same or not cs:
            ok = False
            bad = np.argwhere(np.abs(ref - img).max(axis=-1) > 0)
            d = (img - ref)[..., :3].sum(axis=-1)
            print("   differing pixels:", len(bad), "of", ref.shape[0] * ref.shape[1], bad[:5].tolist(), " pool < ref: %d  pool > ref: %d  sum ref %.3f pool %.3f" % ((d < 0).sum(), (d > 0).sum(), float(ref[..., :3].sum()), float(img[..., :3].sum())), flush=True)
            for (y, x) in bad[:6]:
                print("     ", (int(y), int(x)), "ref", ref[y, x].tolist(), "pool", img[y, x].tolist(), flush=True)
            print("   counts ref (%d %d %d) pool (%d %d %d)" % (rc.closest, rc.shadow, rc.shaded, c.closest, c.shadow, c.shaded), flush=True)
            if i >= 1: break
    print("POOL CHECK", "OK" if ok else "FAILED")
    pr.close(); sg.close(); dev.close()

main()
