"""CPU: the hand-over protocol of the pool kernel (loupiote_amd/csrc/pool_kernels.h: rings of record indices behind LDS spin locks, the block's admission
word, room in the TRACE ring reserved before a batch is shaded, the end condition) restated with std::atomic and plain memory and run under ThreadSanitizer — threads as waves (tests/tools/pool_model.cpp).
A data race on a ring slot or a path record, a record in two places, a path finished twice or never, a radiance summed out of order: all fail here,
without a GPU.  (The first version of the kernel read the slots it took AFTER releasing the ring's lock; this model found it.)"""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def model(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("pool") / "pool_model")
    subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=thread", "-pthread", os.path.join(ROOT, "tests", "tools", "pool_model.cpp"), "-o", exe])
    return exe


# blocks, waves, records, paths, shader-first waves, refill threshold, TRACE payload slots
@pytest.mark.parametrize("cfg", ["2 6 256 12000 2 44 128", "1 8 256 6000 0 20 128", "2 4 512 8000 4 63 256", "1 3 256 3000 1 0 128", "1 16 1024 20000 2 44 1024",
                                 "1 2 64 3000 1 44 128", "1 8 2048 12000 2 44 128"])
def test_pool_protocol_is_race_free_and_conserves_paths(model, cfg):
    p = subprocess.run([model] + cfg.split(), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-3000:]
    assert "ThreadSanitizer" not in p.stdout, p.stdout[-3000:]
    assert p.stdout.strip().endswith("OK"), p.stdout[-500:]


def test_kernel_reads_its_ring_slots_under_the_lock():
    """the property the model enforces, checked on the kernel's text: in pool_pop the slot read sits between the lock and the unlock"""
    src = open(os.path.join(ROOT, "loupiote_amd", "csrc", "pool_kernels.h")).read()
    body = src[src.index("uint32_t pool_pop("):src.index("uint32_t pool_count(")]
    assert body.index("pool_lock(") < body.index("idx = rbuf[") < body.index("pool_unlock(")
    body = src[src.index("uint32_t pool_trace_pop("):src.index("uint32_t pool_admit(")]
    assert body.index("pool_lock(") < body.index("a = slot[0]") < body.index("pool_unlock(")
