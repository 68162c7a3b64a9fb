"""-m gpu: the multi-GPU frame exchange behind the C ABI (include/lpt.h "multi-GPU frame exchange").

N ranks are emulated on the one GPU of the box by N sharded renderers of one process
(lpt_renderer_exchange_local: the RCCL send / recv pairs become device copies, everything else — tile
ownership, packing, the staging layout, the unpack on rank 0, the presented frame — is the code the
RCCL path runs).  The RCCL calls themselves are driven through a one-rank communicator in a child
process that never imports torch (tests/tools/comm_one_rank.py)."""
import os
import subprocess
import sys

import numpy as np
import pytest

import loupiote_amd as lp
from loupiote_amd import testing as T

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("pipeline")]   # every test body over the three arms of tests/conftest.py PIPELINES: k_path, the per-bounce launches, the shipped defaults
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _setup(device, glb):
    scene = lp.Scene()
    lp.loaders.load_gltf(glb, scene)
    scene.set_light(0, T.cornell_light())
    sg = lp.SceneGPU.new_from_scene(scene, device)
    pr = lp.ProbeGPU(device, T.CORNELL_PROBE, 1, 1)
    return sg, pr


def _renderer(device, sg, pr, w, h, bounces, rank=0, world=1, tile=(32, 8), weights=None, options=None):
    r = lp.Renderer(device, (w, h))
    r.downsample_factor = 1.0
    r.resize(device, sg, pr, (w, h))
    r.set_max_bounces(bounces)
    r.set_vfov(T.VFOV)
    for k, v in (options or {}).items():
        r.set_option(k, v)
    if world > 1:
        r.set_shard(rank, world, tile[0], tile[1], weights=weights)
        r.set_resources(device, sg, pr)
    r.reset_accumulation()
    r.accumulate = True
    return r


@pytest.mark.parametrize("w,h,world,tile", [(200, 120, 2, (32, 8)), (200, 120, 3, (32, 8)), (203, 117, 2, (8, 8)), (97, 61, 5, (16, 8)), (64, 64, 8, (32, 8)),
                                            (203, 117, 3, (64, 2)), (97, 61, 2, (16, 12)),   # tile sides that are not multiples of 8 (row-major inside a tile)
                                            (203, 117, 40, (8, 8)), (97, 61, 70, (8, 8))])   # unit weights, more ranks than the weighted rule's tables hold (32 / 64)
def test_progressive_frames_exchanged_every_frame_equal_one_gpu(device, cornell_glb, w, h, world, tile):
    """ADVICE r1 (dist.py:38): exchange after EVERY progressive frame; rank 0's presented frame must equal the single-GPU
    frame each time (an in-place reduce into rank 0's accumulation buffer double-counts from the second frame on).
    Ragged sizes with 8x8 tiles exercise raygen blocks that are only partly inside the slot range (ADVICE device.hip:834)."""
    sg, pr = _setup(device, cornell_glb)
    view = T.look(T.CORNELL_EYE, T.CORNELL_DIR)
    one = _renderer(device, sg, pr, w, h, 4)
    ranks = [_renderer(device, sg, pr, w, h, 4, q, world, tile) for q in range(world)]
    for frame in range(3):
        one.raytrace(view)
        want = one.read_radiance()
        want8 = one.read_pixels()
        for r in ranks:
            r.raytrace(view)
        ranks[0].exchange_local(ranks[1:])
        got = ranks[0].read_radiance()
        assert got.tobytes() == want.tobytes(), "frame %d" % frame
        assert ranks[0].read_pixels().tobytes() == want8.tobytes()
        # the other ranks still present their own tiles only
        part = ranks[1].read_radiance()
        assert np.all(part[..., 3][want[..., 3] > 0] <= 1.0) and part[..., 3].sum() < want[..., 3].sum()
    # ray totals: the shards partition the work
    tot = sum(r.ray_counts().closest for r in ranks)
    assert tot == one.ray_counts().closest
    for r in ranks + [one]:
        r.close()
    pr.close()
    sg.close()


@pytest.mark.parametrize("mode,world,tile", [(lp.BlitMode.DenoisedPathrace, 2, (32, 8)), (lp.BlitMode.Temporal, 3, (32, 8)), (lp.BlitMode.DenoisedPathrace, 5, (16, 8))])
def test_denoising_modes_exchange_owned_tiles_of_the_filter_inputs(device, cornell_glb, mode, world, tile):
    """config 5 on N GPUs: every rank traces its tiles, the owned pixels of the three filter inputs (40 B per pixel) travel
    to rank 0, rank 0 filters the whole frame — bit-identical to the single-GPU denoiser frame after frame, with a moving
    camera so that reprojection crosses tile borders (SPEC §15.5)"""
    sg, pr = _setup(device, cornell_glb)
    W, H = 203, 117
    one = _renderer(device, sg, pr, W, H, 3)
    ranks = [_renderer(device, sg, pr, W, H, 3, q, world, tile) for q in range(world)]
    for r in [one] + ranks:
        r.set_blit_mode(mode)
        r.reset_accumulation()
    for f in range(4):
        view = T.look((0.15 * f, 0.6 + 0.05 * f, 13.5 - 0.2 * f), T.CORNELL_DIR)
        one.raytrace(view)
        for r in ranks:
            r.raytrace(view)
        ranks[0].exchange_local(ranks[1:])
        assert ranks[0].read_radiance().tobytes() == one.read_radiance().tobytes(), "frame %d" % f
        for got, want in zip(ranks[0].read_denoiser(), one.read_denoiser()):
            assert got.tobytes() == want.tobytes()
    with pytest.raises(lp.Error):
        ranks[0].exchange_local(ranks[1:])           # the inputs of this frame have been consumed
    for r in ranks + [one]:
        r.close()
    pr.close()
    sg.close()


@pytest.mark.parametrize("w,h,tile,weights,mode", [
    (200, 120, (32, 8), (1, 3, 2), lp.BlitMode.Pahtrace),
    (203, 117, (8, 8), (5, 8, 8, 8), lp.BlitMode.Pahtrace),
    (97, 61, (16, 8), (0, 2, 2, 1), lp.BlitMode.Pahtrace),                # rank 0 owns nothing: a pure compositor
    (203, 117, (32, 8), (2, 8, 7), lp.BlitMode.DenoisedPathrace),
    (64, 64, (32, 8), (3, 8, 8, 8, 8, 8, 8, 8), lp.BlitMode.Temporal),
])
def test_weighted_tile_ownership_does_not_change_a_bit(device, cornell_glb, w, h, tile, weights, mode):
    """lpt_renderer_set_shard_weighted: ranks own tiles in proportion to their weights (rank 0, which also unpacks, reads back or
    filters the frame, gets fewer) — the RNG is keyed by the global pixel, so the exchanged frame equals the single-GPU frame bit
    for bit, frame after frame, path tracing and denoising; the ranks' work follows the weights"""
    sg, pr = _setup(device, cornell_glb)
    world = len(weights)
    one = _renderer(device, sg, pr, w, h, 4)
    ranks = [_renderer(device, sg, pr, w, h, 4, q, world, tile, weights) for q in range(world)]
    den = mode != lp.BlitMode.Pahtrace
    for r in [one] + ranks:
        r.set_blit_mode(mode)
        r.reset_accumulation()
        r.accumulate = not den
        r.reset_ray_counts()
    for f in range(3):
        view = T.look((0.15 * f, 0.6, 13.5 - 0.2 * f), T.CORNELL_DIR) if den else T.look(T.CORNELL_EYE, T.CORNELL_DIR)
        one.raytrace(view)
        for r in ranks:
            r.raytrace(view)
        ranks[0].exchange_local(ranks[1:])
        assert ranks[0].read_radiance().tobytes() == one.read_radiance().tobytes(), "frame %d" % f
        if den:
            for got, want in zip(ranks[0].read_denoiser(), one.read_denoiser()):
                assert got.tobytes() == want.tobytes()
    counts = [r.ray_counts().closest for r in ranks]
    assert sum(counts) == one.ray_counts().closest
    assert all((c == 0) == (wq == 0) for c, wq in zip(counts, weights))
    # a peer with other weights does not complete the shard set
    other = _renderer(device, sg, pr, w, h, 4, 1, world, tile, None)
    with pytest.raises(lp.Error):
        ranks[0].exchange_local([other] + ranks[2:])
    other.close()
    for r in ranks + [one]:
        r.close()
    pr.close()
    sg.close()


def test_exchange_local_rejects_incomplete_shard_sets(device, cornell_glb):
    sg, pr = _setup(device, cornell_glb)
    a = _renderer(device, sg, pr, 64, 64, 2, 0, 3)
    b = _renderer(device, sg, pr, 64, 64, 2, 1, 3)
    with pytest.raises(lp.Error) as e:
        a.exchange_local([b])           # rank 2 is missing
    assert e.value.kind == "InvalidArg"
    with pytest.raises(lp.Error):
        a.exchange_local([b, b])        # rank 1 twice
    with pytest.raises(lp.Error):
        b.exchange_local([a, a])        # root must be rank 0
    for r in (a, b):
        r.close()
    pr.close()
    sg.close()


def test_exchange_without_a_communicator_is_a_noop(device, cornell_glb):
    sg, pr = _setup(device, cornell_glb)
    r = _renderer(device, sg, pr, 96, 64, 3)
    view = T.look(T.CORNELL_EYE, T.CORNELL_DIR)
    r.raytrace(view)
    a = r.read_radiance()
    r.exchange()
    r.exchange(lp.EXCHANGE_REDUCE)
    assert r.read_radiance().tobytes() == a.tobytes()
    with pytest.raises(lp.Error):
        r.exchange(7)
    r.close()
    pr.close()
    sg.close()


def test_rccl_one_rank_communicator_without_torch():
    """lpt_comm_unique_id / lpt_comm_create / lpt_renderer_set_comm / lpt_renderer_exchange (both modes, and the denoiser
    inputs) through ctypes in a process that never imports torch: the library links librccl itself."""
    env = dict(os.environ, PYTHONPATH=ROOT)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "tools", "comm_one_rank.py")], env=env, capture_output=True, text=True, timeout=600)
    print(p.stdout[-3000:], p.stderr[-3000:])
    assert p.returncode == 0
    assert "COMM_ONE_RANK_OK" in p.stdout


@pytest.mark.parametrize("world,weights", [(2, None), (3, (1, 3, 2))])
def test_sharded_frames_cut_into_runs_of_tiles_equal_one_gpu(device, cornell_glb, monkeypatch, world, weights):
    """a recorded batch too large for one wavefront on a SHARDED frame is cut into runs of the rank's tiles (not tile rows; no
    piecewise read-back): four recorded calls per frame on every emulated rank, cut small, exchanged — the single-GPU frame"""
    sg, pr = _setup(device, cornell_glb)
    view = T.look(T.CORNELL_EYE, T.CORNELL_DIR)
    w, h = 203, 117
    one = _renderer(device, sg, pr, w, h, 4)
    cut = {"wavefront_rays": 6000}          # about five 32x8 tiles of 4 samples per wavefront
    ranks = [_renderer(device, sg, pr, w, h, 4, q, world, (32, 8), weights, options=cut) for q in range(world)]
    for frame in range(2):
        for _ in range(4):
            one.raytrace(view)
            for r in ranks:
                r.raytrace(view)
        assert all(r.submission_stats()[2] == 4 for r in ranks)   # still recorded
        ranks[0].exchange_local(ranks[1:])
        assert ranks[0].read_radiance().tobytes() == one.read_radiance().tobytes(), "frame %d" % frame
    assert ranks[1].submission_stats()[1] > 2 * 3                 # several wavefronts per frame
    assert sum(r.ray_counts().closest for r in ranks) == one.ray_counts().closest
    for r in ranks + [one]:
        r.close()
    pr.close()
    sg.close()


@pytest.mark.parametrize("w,h,world,tile,weights", [(200, 120, 2, (32, 8), None), (203, 117, 3, (8, 8), None), (97, 61, 5, (16, 8), (1, 3, 2, 0, 2)), (203, 117, 8, (32, 8), None),
                                                     (203, 117, 3, (64, 2), None)])
def test_host_side_gather_every_rank_writes_its_own_pixels_into_one_frame(device, cornell_glb, w, h, world, tile, weights, tmp_path):
    """the host-side gather: when the consumer of the frame is the host, every rank writes its OWNED pixels of the mean radiance straight into ONE
    whole-frame destination in page-locked host memory (lpt_renderer_read_radiance_owned) — no exchange on the GPUs, each GPU's 1/N over its own link.  The
    emulated ranks fill a frame that starts as NaN; it must equal the single-GPU read_radiance bit for bit, through lpt_host_alloc memory and through a
    memory-mapped file registered with lpt_host_register (the shared-memory segment of a multi-process host)."""
    sg, pr = _setup(device, cornell_glb)
    view = T.look(T.CORNELL_EYE, T.CORNELL_DIR)
    one = _renderer(device, sg, pr, w, h, 4)
    ranks = [_renderer(device, sg, pr, w, h, 4, q, world, tile, weights) for q in range(world)]
    frame = lp.pinned_array((h, w, 4))
    shared = np.memmap(str(tmp_path / "frame.bin"), dtype=np.float32, mode="w+", shape=(h, w, 4))
    lp.host_register(shared)
    for k in range(2):
        for _ in range(2):
            one.raytrace(view)
            for r in ranks:
                r.raytrace(view)
        want = one.read_radiance()
        for dst in (frame, shared):
            dst[...] = np.nan
            for r in ranks:
                r.read_radiance_owned(dst)
            assert np.asarray(dst).tobytes() == want.tobytes(), (k, type(dst))
    # a rank alone fills exactly its own pixels
    frame[...] = np.nan
    ranks[0].read_radiance_owned(frame)
    filled = ~np.isnan(frame[..., 0])
    assert filled.sum() > 0 and (world == 1 or not filled.all())
    with pytest.raises(lp.Error):
        ranks[0].read_radiance_owned(np.empty((h, w, 4), np.float32))     # pageable memory: refused, not silently staged
    lp.host_unregister(shared)
    for r in ranks + [one]:
        r.close()
    pr.close()
    sg.close()
