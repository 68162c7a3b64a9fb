"""Child process of tests/test_gpu_exchange.py: a ONE-rank RCCL communicator driven through the C ABI only.
torch is never imported here — the frame exchange is plain RCCL inside libloupiote_hip.so."""
import os
import sys

import numpy as np

import loupiote_amd as lp
from loupiote_amd import testing as T

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    dev = lp.Device(0)
    glb = open(os.path.join(ROOT, "tests", "golden", "cornell-box.glb"), "rb").read()
    scene = lp.Scene()
    lp.loaders.load_gltf(glb, scene)
    scene.set_light(0, T.cornell_light())
    sg = lp.SceneGPU.new_from_scene(scene, dev)
    pr = lp.ProbeGPU(dev, T.CORNELL_PROBE, 1, 1)
    view = T.look(T.CORNELL_EYE, T.CORNELL_DIR)
    W, H = 203, 117

    def mk():
        r = lp.Renderer(dev, (W, H))
        r.downsample_factor = 1.0
        r.resize(dev, sg, pr, (W, H))
        r.set_max_bounces(4)
        r.set_vfov(T.VFOV)
        r.reset_accumulation()
        r.accumulate = True
        return r

    uid = lp.Comm.unique_id()
    assert len(uid) == 128
    comm = lp.Comm(dev, uid, 0, 1)
    assert comm.info() == (0, 1)
    plain, shared = mk(), mk()
    shared.set_comm(comm)
    shared.set_resources(dev, sg, pr)
    shared.reset_accumulation()
    shared.accumulate = True
    for frame in range(3):
        plain.raytrace(view)
        shared.raytrace(view)
        want = plain.read_radiance()
        for mode in (lp.EXCHANGE_GATHER_TILES, lp.EXCHANGE_REDUCE):
            shared.exchange(mode)
            got = shared.read_radiance()
            assert got.tobytes() == want.tobytes(), (frame, mode)
    # denoising BlitMode: the exchange carries the filter inputs and rank 0 filters
    for r in (plain, shared):
        r.set_blit_mode(lp.BlitMode.DenoisedPathrace)
        r.reset_accumulation()
    for frame in range(4):
        plain.raytrace(view)
        shared.raytrace(view)
        shared.exchange(lp.EXCHANGE_GATHER_TILES if frame % 2 == 0 else lp.EXCHANGE_REDUCE)
        assert shared.read_radiance().tobytes() == plain.read_radiance().tobytes(), frame
        for got, want in zip(shared.read_denoiser(), plain.read_denoiser()):
            assert got.tobytes() == want.tobytes()
    assert "torch" not in sys.modules
    # the single-thread / several-communicators form: creation and the exchange inside a group (ncclGroupStart / End)
    from loupiote_amd import _abi as A
    L = A.lib()
    shared.set_blit_mode(lp.BlitMode.Pahtrace)
    plain.set_blit_mode(lp.BlitMode.Pahtrace)
    shared.set_comm(None)
    assert L.lpt_comm_group_begin() == 0
    comm2 = lp.Comm(dev, lp.Comm.unique_id(), 0, 1)
    assert L.lpt_comm_group_end() == 0
    shared.set_comm(comm2)
    shared.set_resources(dev, sg, pr)
    for r in (plain, shared):
        r.reset_accumulation()
        r.accumulate = True
    for frame in range(2):
        plain.raytrace(view)
        shared.raytrace(view)
        assert L.lpt_comm_group_begin() == 0
        shared.exchange(lp.EXCHANGE_GATHER_TILES)
        assert L.lpt_comm_group_end() == 0
        assert shared.read_radiance().tobytes() == plain.read_radiance().tobytes(), frame
    # two communicators driven by ONE thread, both exchanges of a frame inside one group bracket (the single-process /
    # several-GPU host of INTEGRATION.md): the RCCL operations are issued by the outermost lpt_comm_group_end, which then
    # enqueues what consumes them (unpack, filter passes) for EVERY renderer of the bracket — path tracing and denoising
    assert L.lpt_comm_group_begin() == 0
    comm3 = lp.Comm(dev, lp.Comm.unique_id(), 0, 1)
    assert L.lpt_comm_group_end() == 0
    view2 = T.look((0.4, 0.5, 12.5), (-0.03, 0.0, -1.0))
    for mode in (lp.BlitMode.Pahtrace, lp.BlitMode.DenoisedPathrace):
        pa, pb, sa, sb = mk(), mk(), mk(), mk()          # fresh renderers: the ASVGF history starts empty on all four
        sa.set_comm(comm2); sa.set_resources(dev, sg, pr)
        sb.set_comm(comm3); sb.set_resources(dev, sg, pr)
        for r in (pa, pb, sa, sb):
            r.set_blit_mode(mode)
            r.reset_accumulation()
            r.accumulate = mode == lp.BlitMode.Pahtrace
        for frame in range(3):
            pa.raytrace(view); sa.raytrace(view)
            pb.raytrace(view2); sb.raytrace(view2)
            assert L.lpt_comm_group_begin() == 0
            assert L.lpt_comm_group_begin() == 0          # brackets nest: only the outermost end issues
            sa.exchange(lp.EXCHANGE_GATHER_TILES)
            assert L.lpt_comm_group_end() == 0
            sb.exchange(lp.EXCHANGE_REDUCE if frame % 2 else lp.EXCHANGE_GATHER_TILES)
            assert L.lpt_comm_group_end() == 0
            assert sa.read_radiance().tobytes() == pa.read_radiance().tobytes(), (mode, frame)
            assert sb.read_radiance().tobytes() == pb.read_radiance().tobytes(), (mode, frame)
        for r in (sa, sb):
            r.set_comm(None)
        for r in (pa, pb, sa, sb):
            r.close()
    assert L.lpt_comm_group_end() != 0                    # an end without a begin is an error, not a crash
    comm3.close()
    shared.set_comm(None)
    comm2.close()
    for r in (plain, shared):
        r.close()
    comm.close()
    pr.close()
    sg.close()
    dev.close()
    print("COMM_ONE_RANK_OK")


if __name__ == "__main__":
    main()
