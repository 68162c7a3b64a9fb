"""Run by tests/test_gpu_denoiser.py in a subprocess: one-rank RCCL group on the GPU, a sharded (world 2, rank 0) renderer
in a denoising mode, `dist.exchange_denoiser_inputs` (three reduces through torch.distributed on the renderer's stream)
followed by the filter passes, and `OwnedTileGather` / the dense reduce on a device buffer — the multi-GPU plumbing with
real device pointers, as far as one GPU can exercise it."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import loupiote_amd as lp  # noqa: E402
from loupiote_amd import dist as D, testing as T  # noqa: E402

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29541")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
dev = lp.Device(0)
scene = lp.Scene()
lp.loaders.load_gltf(open(os.path.join(ROOT, "tests", "golden", "cornell-box.glb"), "rb").read(), scene)
scene.set_light(0, T.cornell_light())
sg = lp.SceneGPU.new_from_scene(scene, dev)
pr = lp.ProbeGPU(dev, T.CORNELL_PROBE, 1, 1)
W, H = 96, 64
r = lp.Renderer(dev, (W, H))
r.downsample_factor = 1.0
r.resize(dev, sg, pr, (W, H))
r.set_max_bounces(3)
r.set_vfov(T.VFOV)
r.set_shard(0, 2)
r.set_resources(dev, sg, pr)
r.set_blit_mode(lp.BlitMode.Temporal)
ext = torch.cuda.ExternalStream(r.stream(), device=torch.device("cuda", 0))
view = T.look(T.CORNELL_EYE, T.CORNELL_DIR)
for _ in range(3):
    r.raytrace(view)
    D.exchange_denoiser_inputs(r, torch.device("cuda", 0), dst=0, stream=ext)
out = r.read_radiance()
g, m, rad, hist = r.read_denoiser()
own = D.owned_mask(W, H, 0, 2)
assert np.all(np.isfinite(out)) and hist[own].max() == 3 and hist[~own].max() <= 3
assert np.any(g[own, 0] != 0xFFFFFFFF) and not g[~own].any()          # rank 0's tiles only: the other rank's stay zero
# frame exchange on a device buffer: dense reduce and compact gather leave a one-rank frame unchanged
ptr, nbytes = r.radiance_device_ptr()
buf = torch.as_tensor(D.DevView(ptr, nbytes // 4, "<f4"), device=torch.device("cuda", 0))
before = buf.clone()
with torch.cuda.stream(ext):
    D.reduce_radiance(buf, dst=0)
    D.OwnedTileGather(W, H, 0, 1, device=torch.device("cuda", 0))(buf.view(H, W, 4))
torch.cuda.synchronize()
assert torch.equal(buf, before)
r.close(); pr.close(); sg.close(); dev.close()
dist.destroy_process_group()
print("dist-den-check ok")
