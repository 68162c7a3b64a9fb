"""experiment: per node of the packet walk, how many children pass the packet's own test, how many are entered (library variant built with -DLPT_EXP_PACKET_COUNT)"""
import sys
import loupiote_amd as lp
from loupiote_amd import scenes, testing as T
dev = lp.Device()
desc = scenes.synthetic_atrium()
sg = lp.SceneGPU.new_from_scene(scenes.to_product(desc), dev)
pr = lp.ProbeGPU(dev, desc["probe"], desc["probe"].shape[1], desc["probe"].shape[0])
view = T.look(desc["camera"]["origin"], desc["camera"]["direction"])
for size in ((1920, 1080), (1280, 720), (3840, 2160)):
    r = lp.Renderer(dev, size)
    r.downsample_factor = 1.0
    r.resize(dev, sg, pr, size)
    r.set_max_bounces(1)
    r.set_vfov(T.VFOV)
    r.set_option("packet_primary", 1)
    r.set_option("path_rays", 0)
    r.enable_stats(True)
    r.reset_accumulation(); r.accumulate = True; r.reset_ray_counts()
    r.raytrace_n(view, 4)
    r.synchronize()
    c = r.ray_counts()
    nodes = c.packet_nodes
    print(size, "packets", c.primary // 64, "nodes/packet %.2f tris/packet %.2f" % (nodes / (c.primary / 64), c.packet_tris / (c.primary / 64)),
          "| per node: candidates %.2f entered %.2f coherent %.3f" % (c.occluder_cache_found / nodes, c.occluder_cache_hits / nodes, c.shadow_occluded / nodes))
    r.close()
