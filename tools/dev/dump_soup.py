"""dev: the bench scene's world-space triangle soup as tests/tools/bvh_check's input (offline BVH-quality runs on the CPU)
usage: python tools/dev/dump_soup.py out.bin [atrium|atrium_quads|hall|helmet]"""
import struct
import sys

import numpy as np

sys.path.insert(0, ".")
from loupiote_amd import scenes

which = sys.argv[2] if len(sys.argv) > 2 else "atrium"
desc = {"atrium": lambda: scenes.synthetic_atrium(textures=False), "atrium_quads": lambda: scenes.synthetic_atrium(textures=False, shell_quads=True),
        "hall": scenes.synthetic_hall, "helmet": lambda: scenes.synthetic_helmet(textures=False)}[which]()
tris = []
for blas, m, _mat in desc["instances"]:
    mesh = desc["meshes"][blas - 1]                      # BLAS index in the final scene: the dummy entry 0 precedes
    M = np.asarray(m, np.float32).reshape(4, 4)          # column-major: columns are stored one after the other
    p = np.asarray(mesh["positions"], np.float32).reshape(-1, 3)
    w = p @ M[:3, :3] + M[3, :3]
    idx = np.asarray(mesh["indices"]).reshape(-1, 3)
    tris.append(w[idx].reshape(-1, 9))
tris = np.concatenate(tris).astype("<f4")
with open(sys.argv[1], "wb") as f:
    f.write(struct.pack("<I", len(tris)))
    f.write(tris.tobytes())
print(len(tris), "triangles", tris.reshape(-1, 3).min(0), tris.reshape(-1, 3).max(0))
