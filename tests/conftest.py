import os
import sys

import pytest

# PyTorch-ROCm bundles its own libamdhip64.so.7; the loader keeps whichever copy is loaded first for the whole
# process.  Tests that hand device pointers to torch (exchange plumbing) need torch's copy to be that one — the
# order bench.py has anyway — so torch is imported before libloupiote_hip.so.
try:
    import torch  # noqa: F401
except Exception:  # pragma: no cover - torch is optional for the pure C-ABI tests
    torch = None

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # a fresh checkout has no built artefacts (they are git-ignored): build the library once (hipcc cross-compiles
    # gfx950 without a GPU; `make` is a no-op when it is up to date).  The oracle builds itself on import.
    so = os.path.join(ROOT, "loupiote_amd", "libloupiote_hip.so")
    if not os.path.exists(so):
        from loupiote_amd import build
        build.build()


@pytest.fixture(scope="session")
def cornell_glb():
    with open(os.path.join(ROOT, "tests", "golden", "cornell-box.glb"), "rb") as f:
        return f.read()


@pytest.fixture(scope="session")
def device():
    import loupiote_amd as lp
    dev = lp.Device(0)
    yield dev
    dev.close()


PIPELINES = {
    # every bounce behind the primary hits in ONE launch, a lane carries a path (k_path), whatever the size of the wavefront; bounce 0 through the packet
    # kernel at every resolution (the default picks it by pixel footprint: not for the small frames most tests render)
    "path": {"path_rays": 0x7FFFFFFF, "packet_primary": 1, "coop_rays": 0},
    # the per-bounce launches of renderer.rs:484-509 (k_shade + k_trace) with a step budget of 16 (default 48), so that in every parity test a good part of the
    # rays is finished by the wave-cooperative kernel (k_trace_coop) and the rest by the per-lane kernel; the packet choice is the library's.  tail_lanes 0: the
    # shipped form of these launches — tails finished in place instead of a budget — runs in the "default" arm (where the frame takes the per-bounce launches) and
    # in tests/test_gpu_tail.py
    "per_bounce": {"path_rays": 0, "step_budget": 16, "tail_lanes": 0, "coop_rays": 0},
    # the configuration as shipped: nothing forced (kernel choice by ray count and pixel footprint — a wave per ray for the tiny frames many tests render, LPT_OPT_COOP_RAYS —,
    # tails in place, quad packets) — ADVICE r04.  The other arms switch the tiny-wavefront rule off (coop_rays 0) so that they exercise the kernel they name at every size
    "default": {},
}


@pytest.fixture(params=list(PIPELINES))
def pipeline(request, monkeypatch):
    """Runs a test body over every form of the frame pipeline (PIPELINES above) and over the shipped defaults.  The library picks between the forms by
    the wavefront's ray count (LPT_OPT_COOP_RAYS, LPT_OPT_PATH_RAYS); all of them must give the oracle's frame at every size.  Modules opt in with
    `pytestmark = pytest.mark.usefixtures("pipeline")`."""
    from loupiote_amd import api
    # the shipped defaults pick k_path for the small frames most tests render — what the "path" arm already forces; the arm earns its time where the
    # selection logic itself is exercised: the full-size vectors and configs (packets by pixel footprint, quad packets, the spatial cut, the cross-overs)
    if request.param == "default" and request.module.__name__.rsplit(".", 1)[-1] not in ("test_gpu_golden", "test_gpu_configs", "test_gpu_atrium", "test_gpu_deferred"):
        pytest.skip("the `default` arm runs with the full-size vectors / configs / atrium / deferred modules")
    monkeypatch.setattr(api, "DEFAULT_OPTIONS", dict(PIPELINES[request.param]))
    return request.param
