"""dev: configs 2 / 3 of tools/configs_timing.py alone, with more warm-up frames (argv[1]) — to separate first-use allocations from steady state"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import configs_timing as ct
import loupiote_amd as lp
from loupiote_amd import scenes, testing as T

warm = int(sys.argv[1]) if len(sys.argv) > 1 else 3
dev = lp.Device(0)
glb = open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests", "golden", "cornell-box.glb"), "rb").read()
s = lp.Scene(); lp.loaders.load_gltf(glb, s); s.set_light(0, T.cornell_light())
ct.measure(dev, "2 warm=%d" % warm, s, T.CORNELL_PROBE, (T.CORNELL_EYE, T.CORNELL_DIR), 1024, 1024, 4, 8, warm=warm)
d = scenes.synthetic_helmet()
ct.measure(dev, "3 warm=%d" % warm, scenes.to_product(d), d["probe"], (d["camera"]["origin"], d["camera"]["direction"]), 1920, 1080, 8, 8, warm=warm)
