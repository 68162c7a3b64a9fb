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
    for flag in ("--gpus", "--steps", "--warmup", "--exchange", "--pipeline", "--emulate-shard", "--spawn-dry-run", "--gltf", "--probe", "--camera"):
        assert flag in p.stdout
    sys.path.insert(0, ROOT)
    import bench
    a = bench.parse_args(["--gltf", "a.glb", "--gltf", "b.glb", "--probe", "e.hdr", "--camera", "0,1,2,0,0,-1"])
    assert a.gltf == ["a.glb", "b.glb"] and a.probe == "e.hdr" and a.exchange == "auto"     # real assets, when they appear, take the same span; N>1 defaults to auto


def test_bench_starts_its_own_ranks_and_they_rendezvous():
    """`python bench.py --gpus 2` (the driver's command shape, no launcher): the script starts two ranks of itself before
    anything touches a GPU; with --spawn-dry-run they only meet over gloo and rank 0's line is relayed"""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--spawn-dry-run"], capture_output=True, text=True, timeout=600, env=env)
    assert p.returncode == 0, p.stderr[-3000:]
    line = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    assert line["dry_run"] is True and line["n_gpus"] == 2 and line["ranks_seen"] == [0, 1]


def test_bench_under_an_external_launcher_still_is_one_rank():
    """python -m torch.distributed.run --nproc-per-node 2 bench.py --gpus 2: WORLD_SIZE is set, nothing is re-spawned"""
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", "29533",
                        os.path.join(ROOT, "bench.py"), "--gpus", "2", "--spawn-dry-run"], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [json.loads(l) for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1 and lines[0]["ranks_seen"] == [0, 1]


@pytest.mark.skipif(os.path.exists("/dev/kfd"), reason="a GPU is present")
def test_bench_refuses_more_ranks_than_gpus_loudly():
    """on a box with fewer GPUs than --gpus the launcher says so and exits non-zero at once (no hang, no JSON line)"""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    t0 = __import__("time").time()
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"], capture_output=True, text=True, timeout=600, env=env)
    assert p.returncode != 0 and "GPU(s) visible" in p.stderr
    assert not any(l.startswith("{") for l in p.stdout.splitlines())
    assert __import__("time").time() - t0 < 300


def test_a_dying_rank_takes_the_job_down_with_its_stderr(tmp_path):
    """a rank that dies makes the launcher stop the others and exit non-zero with that rank's stderr"""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["LPT_BENCH_TEST_DIE_RANK"] = "1"
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--spawn-dry-run"], capture_output=True, text=True, timeout=600, env=env)
    assert p.returncode != 0
    assert "rank 1 exited" in p.stderr and "LPT_BENCH_TEST_DIE_RANK" in p.stderr


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


def test_replayed_profile_figures_live_under_one_key_and_frac_fabric_rides_with_frac():
    """what bench.py loads from profiles/*.json (counter passes of an earlier run) sits under roofline.from_profiles and is
    labelled as replayed; roofline.frac_fabric = fabric bytes per launch / THIS run's launch time / 8 TB/s stands beside roofline.frac"""
    sys.path.insert(0, ROOT)
    import bench
    obj, traffic = bench.from_profiles_object(1.0)
    assert set(obj) == {"what", "source", "traffic", "traffic_unit", "limits", "binding_limit", "kernel_source_hash", "stale"} and "NOT measured in this run" in obj["what"]
    # round 5: the profiles record the device code they were measured on; a changed kernel with un-refreshed profiles is flagged
    assert obj["kernel_source_hash"]["this_tree"] == bench.kernel_source_hash() and len(obj["kernel_source_hash"]["this_tree"]) == 16
    assert obj["stale"] is (obj["kernel_source_hash"]["profiles"] != obj["kernel_source_hash"]["this_tree"])
    assert traffic == obj["traffic"] and traffic > 1e9 and obj["source"]["traffic"].startswith("profiles/traffic.json")
    assert obj["binding_limit"]["name"] in ("valu_issue", "vector_memory_path") and 0 < obj["binding_limit"]["frac"] < 1
    frac, frac_fabric = bench.roofline_fractions(7990.0, traffic, 1.008)
    assert abs(frac - 7990.0 / 8000.0) < 1e-12
    assert abs(frac_fabric - traffic / 1.008e-3 / 1e9 / 8000.0) < 1e-12 and 0.2 < frac_fabric < 0.6
    assert bench.roofline_fractions(7990.0, None, 1.0)[1] is None
    # the keys of the roofline object the line carries, as the source states them
    src = open(os.path.join(ROOT, "bench.py")).read()
    for key in ('"frac": rf_frac', '"frac_fabric": rf_frac_fabric', '"from_profiles": profiles_j', '"traffic": fabric_bytes', '"shard_emulation": shard_emulation',
                '"frac_fetched": rf_frac_fetched', '"Grays_per_s": rays_per_launch', '"fixed_counts": fixed', 'out["exchange_forms"] = exchange_forms', '"stage_ms_per_rank": stage_ms_ranks'):
        assert key in src, key
    # the fixed per-config counts of SURVEY 8d are committed with the fixtures and are what `frac` is computed from
    fixed = bench.fixed_traversal_counts()
    assert fixed is None or (10 < fixed["nodes_per_ray"] < 16 and 3 < fixed["tris_per_ray"] < 6 and 6 < fixed["shadow_nodes_per_ray"] < 12 and 1 < fixed["shadow_tris_per_ray"] < 3)
    assert '"limits": limits_j,' not in src.split('"roofline": {')[1].split('"from_profiles": profiles_j')[0]   # not beside the live figures any more
