"""CPU: bench.py and the driver entry points import and parse on a machine without a GPU (the measurement itself needs one:
the library fails loudly, there is no CPU path to fall back to)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_help_and_contract_flags():
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--help"], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    for flag in ("--gpus", "--steps", "--warmup", "--exchange", "--pipeline", "--emulate-shard"):
        assert flag in p.stdout


@pytest.mark.skipif(os.path.exists("/dev/kfd"), reason="a GPU is present")
def test_bench_fails_loudly_without_a_gpu():
    """no silent CPU fallback: without a device the bench dies with the library's error, it does not print a JSON line"""
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "0"], capture_output=True, text=True, timeout=600)
    assert p.returncode != 0
    assert not any(l.startswith("{") for l in p.stdout.splitlines())


def test_metric_string_is_baseline_jsons():
    sys.path.insert(0, ROOT)
    want = json.load(open(os.path.join(ROOT, "BASELINE.json")))["metric"]
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert "baseline_metric()" in src and "BASELINE.json" in src
    assert "Mrays/s" in want


def test_graft_entry_has_build_and_smoke():
    sys.path.insert(0, ROOT)
    import __graft_entry__ as g
    assert callable(g.build) and callable(g.smoke)
