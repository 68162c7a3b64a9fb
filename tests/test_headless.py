"""The C++ mirror (include/loupiote.hpp) and the headless driver (examples/headless.cpp) that replays the
standalone app's per-frame protocol (reference crates/standalone/src/app.rs:297-318, save_screenshot :172-187)."""
import os
import subprocess

import numpy as np
import pytest

import loupiote_amd as lp
from loupiote_amd import _abi as A

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _build(tmp_path):
    exe = str(tmp_path / "headless")
    subprocess.check_call(["g++", "-std=c++17", "-Wall", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", "headless.cpp"),
                           "-L" + os.path.join(ROOT, "loupiote_amd"), "-lloupiote_hip", "-Wl,-rpath," + os.path.join(ROOT, "loupiote_amd"), "-o", exe])
    return exe


@pytest.mark.skipif(os.path.exists("/dev/kfd"), reason="a GPU is present")
def test_headless_driver_builds_and_fails_loudly_without_gpu(tmp_path):
    exe = _build(tmp_path)
    p = subprocess.run([exe, os.path.join(ROOT, "tests", "golden", "cornell-box.glb"), str(tmp_path / "o.png")], capture_output=True, text=True)
    assert p.returncode == 1 and "no CPU fallback" in p.stderr


def test_png_writer_round_trips_through_the_loader_decoder(tmp_path):
    from PIL import Image
    rng = np.random.default_rng(4)
    img = rng.integers(0, 256, (300, 257, 4), dtype=np.uint8)   # > 65535 bytes: several stored blocks
    path = str(tmp_path / "a.png")
    assert A.lib().lpt_write_png(path.encode(), A.ptr(img), 257, 300, 257 * 4) == 0
    assert np.array_equal(np.asarray(Image.open(path)), img)
    assert A.lib().lpt_write_png(b"/nonexistent-dir/x.png", A.ptr(img), 257, 300, 257 * 4) == A.LPT_ERR_FILE_NOT_FOUND


@pytest.mark.gpu
def test_headless_driver_renders_and_saves(tmp_path):
    from PIL import Image
    exe = _build(tmp_path)
    out = str(tmp_path / "cornell.png")
    p = subprocess.run([exe, os.path.join(ROOT, "tests", "golden", "cornell-box.glb"), out, "640", "480", "4", "3"], capture_output=True, text=True)
    assert p.returncode == 0, p.stderr
    im = np.asarray(Image.open(out))
    assert im.shape == (240, 320, 4)          # downsample_factor 0.5 like the reference (renderer.rs:225)
    assert '"frames": 4' in p.stdout and im[..., 3].min() == 255


def _build_c(tmp_path):
    exe = str(tmp_path / "multi_gpu")
    subprocess.check_call(["gcc", "-std=gnu11", "-Wall", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", "multi_gpu.c"),
                           "-L" + os.path.join(ROOT, "loupiote_amd"), "-lloupiote_hip", "-Wl,-rpath," + os.path.join(ROOT, "loupiote_amd"), "-lm", "-o", exe])
    return exe


def test_plain_c_multi_gpu_example_compiles_against_the_header(tmp_path):
    """include/lpt.h is a C header: the example is C11, no C++ anywhere on the host side"""
    assert os.path.exists(_build_c(tmp_path))


@pytest.mark.gpu
def test_plain_c_multi_gpu_example_gives_the_same_frame_for_any_rank_count(tmp_path):
    """examples/multi_gpu.c: N ranks inside one process (lpt_renderer_exchange_local), and ONE rank of a one-process RCCL job
    (lpt_comm_unique_id through a file, lpt_comm_create, lpt_renderer_exchange) — the checksum of rank 0's presented frame does
    not depend on N"""
    import json
    exe = _build_c(tmp_path)
    glb = os.path.join(ROOT, "tests", "golden", "cornell-box.glb")
    sums = []
    for n in (1, 2, 3, 8):
        p = subprocess.run([exe, glb, str(n), "203", "117", "3"], capture_output=True, text=True, timeout=300)
        assert p.returncode == 0, p.stderr
        j = json.loads(p.stdout.strip().splitlines()[-1])
        assert j["covered"] == 203 * 117 and j["ranks"] == n
        assert j["host_gather_checksum"] == j["checksum"]      # every rank's own pixels written straight into one host frame: the same frame
        sums.append(j["checksum"])
    # one rank of a "multi-process" job: the RCCL id through a file, and the host-side gather through the shared frame of the C ABI (lpt_host_frame_*)
    env = dict(os.environ, LPT_RANK="0", LPT_WORLD="1", LPT_ID_FILE=str(tmp_path / "rccl.id"), LPT_FRAME_NAME="/lpt_example_%d" % os.getpid())
    p = subprocess.run([exe, glb, "1", "203", "117", "3"], capture_output=True, text=True, timeout=300, env=env)
    assert p.returncode == 0, p.stderr
    j = json.loads(p.stdout.strip().splitlines()[-1])
    assert j["multi_process"] == 1 and os.path.getsize(str(tmp_path / "rccl.id")) == 128
    assert j["host_gather_checksum"] == j["checksum"]          # through lpt_host_frame_create / _ptr / _barrier / _destroy
    assert not os.path.exists("/dev/shm/lpt_example_%d" % os.getpid())
    sums.append(j["checksum"])
    assert len(set(sums)) == 1, sums
