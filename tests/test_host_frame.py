"""CPU: the host-side gather's shared frame and barrier behind the C ABI (lpt_host_frame_*; loupiote_amd/csrc/hostframe.cpp) — POSIX shared memory, a line of
progress words, pause-spinning then futex — between real processes, without a GPU (LPT_HOST_FRAME_HOST_ONLY: the segment is not registered with HIP; on the
GPU box tests/test_gpu_multiproc.py runs the same calls with the GPUs writing into the frame)."""
import multiprocessing as mp
import os
import time

import numpy as np
import pytest

import loupiote_amd as lp


def _name(tag):
    return "/lpt_test_%s_%d_%d" % (tag, os.getpid(), int(time.time() * 1e6) % 1000000)


def _rank(name, rank, world, frames, w, h, q):
    try:
        f = lp.HostFrame.attach(name, w, h, world, host_only=True)
        rows = range(rank, h, world)     # "owned pixels": every world-th row
        for k in range(1, frames + 1):
            for y in rows:
                f.array[y, :, :] = float(k * 1000 + rank)
            f.barrier(rank, 2 * k - 1)   # barrier numbers only ever grow
            # after the barrier the frame is complete on EVERY rank: each row carries this frame's number and its owner
            want = np.array([k * 1000 + (y % world) for y in range(h)], np.float32)
            ok = bool(np.all(f.array[:, 0, 0] == want) and np.all(f.array[:, -1, 3] == want))
            f.barrier(rank, 2 * k)       # nobody overwrites the frame before everybody has checked it
            if not ok:
                q.put((rank, k, "frame incomplete after the barrier"))
                return
        f.close()
        q.put((rank, frames, "ok"))
    except Exception as e:  # noqa: BLE001
        q.put((rank, -1, repr(e)))


def test_shared_frame_between_processes_is_complete_after_every_barrier():
    world, frames, w, h = 3, 200, 64, 48
    name = _name("a")
    f0 = lp.HostFrame.create(name, w, h, world, host_only=True)
    q = mp.get_context("spawn").Queue()
    ps = [mp.get_context("spawn").Process(target=_rank, args=(name, r, world, frames, w, h, q)) for r in range(1, world)]
    for p in ps:
        p.start()
    # rank 0 in this process, on the creator's handle
    res = []
    for k in range(1, frames + 1):
        for y in range(0, h, world):
            f0.array[y, :, :] = float(k * 1000)
        f0.barrier(0, 2 * k - 1)
        want = np.array([k * 1000 + (y % world) for y in range(h)], np.float32)
        assert np.all(f0.array[:, 0, 0] == want), k
        f0.barrier(0, 2 * k)
    for p in ps:
        res.append(q.get(timeout=120))
    for p in ps:
        p.join(60)
    assert sorted(res) == [(r, frames, "ok") for r in range(1, world)], res
    f0.close()
    assert not os.path.exists("/dev/shm" + name)       # the creator unlinks


def test_errors_name_size_world_timeout_and_double_create():
    name = _name("b")
    f = lp.HostFrame.create(name, 16, 8, 2, host_only=True)
    with pytest.raises(lp.Error) as e:
        lp.HostFrame.create(name, 16, 8, 2, host_only=True)          # exists: shm_open(O_EXCL)
    assert e.value.kind == "FileNotFound" or "shm_open" in str(e.value)
    with pytest.raises(lp.Error):
        lp.HostFrame.attach(name, 16, 9, 2, host_only=True)          # another size
    with pytest.raises(lp.Error):
        lp.HostFrame.attach(name, 16, 8, 3, host_only=True)          # another world
    with pytest.raises(lp.Error):
        lp.HostFrame.attach("/lpt_no_such_frame", 16, 8, 2, host_only=True)
    with pytest.raises(lp.Error):
        lp.HostFrame.create("no_slash", 16, 8, 2, host_only=True)
    with pytest.raises(lp.Error):
        f.barrier(2, 1)                                               # rank outside the world
    with pytest.raises(lp.Error):
        f.barrier(0, 0)                                               # frames count from 1
    t0 = time.time()
    with pytest.raises(lp.Error) as e:
        f.barrier(0, 1, timeout_ms=100)                               # rank 1 never arrives
    assert 0.05 < time.time() - t0 < 5.0 and "did not reach frame" in str(e.value)
    f.close()


@pytest.mark.skipif(os.path.exists("/dev/kfd"), reason="a GPU is present")
def test_without_host_only_the_segment_must_be_registered_with_a_device():
    """no CPU fallback: without LPT_HOST_FRAME_HOST_ONLY the frame is a read-back destination of a GPU and needs one"""
    name = _name("c")
    with pytest.raises(lp.Error):
        lp.HostFrame.create(name, 16, 8, 2)
    assert not os.path.exists("/dev/shm" + name)


def _late(name, world, delay_s, q):
    f = lp.HostFrame.attach(name, 8, 8, world, host_only=True)
    time.sleep(delay_s)
    t = time.perf_counter()
    f.barrier(1, 1)
    q.put(time.perf_counter() - t)
    f.close()


def test_a_sleeping_rank_0_is_woken_by_the_late_rank_not_by_a_polling_slice():
    """rank 0 has long stopped spinning when the last rank arrives 50 ms late: the arriving rank sees rank 0's note and wakes it (futex), so the barrier closes
    within a scheduling delay of the arrival — for BOTH sides — not after the rest of a sleep slice"""
    name = _name("d")
    f0 = lp.HostFrame.create(name, 8, 8, 2, host_only=True)
    q = mp.get_context("spawn").Queue()
    p = mp.get_context("spawn").Process(target=_late, args=(name, 2, 0.05, q))
    p.start()
    time.sleep(0.0)
    f0.barrier(0, 1, timeout_ms=20000)      # blocks until the late rank arrives (process start + 50 ms)
    t_done = time.perf_counter()
    late_side = q.get(timeout=60)
    p.join(60)
    assert late_side < 0.02, late_side       # the late rank leaves the barrier within milliseconds of entering it
    f0.close()
